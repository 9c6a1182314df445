"""Small-batch tile plans of the image GEMMs (csrc/gemm3.hip): the fused LSTM launch and the in-loop
backward batch at the per-step row counts of BASELINE configs[3] / [4] at 32 images per GPU (R = 512 / 2048)
and of configs[2] (R = 4096), every plan timed on the same operands.   python tools/small_r_lab.py  (GPU box)
-> gpurun_out/small_r_lab.json"""
import ctypes as C
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlclassification_amd import _lib  # noqa: E402

lib, check = _lib.load(), _lib.check
dev = th.device("cuda:0")


def p4(x):
    return (x + 3) & ~3


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def image(t, k):
    img = th.zeros(lib.marl_image_bytes(t.shape[0], k) + 256, dtype=th.uint8, device=dev)
    check(lib.marl_image_build(t.data_ptr(), t.shape[1], t.shape[0], k, img.data_ptr(), None))
    return img


def padded(t, ld):
    out = th.zeros(t.shape[0], ld, device=dev)
    out[:, : t.shape[1]] = t
    return out


rows = []
for m, n, nin in ((512, 256, 624), (1024, 256, 624)):
    g = th.Generator().manual_seed(m)
    u, h, cprev = th.randn(m, nin, generator=g), th.randn(m, n, generator=g), th.randn(m, n, generator=g)
    wih, whh = th.randn(4 * n, nin, generator=g) / nin ** 0.5, th.randn(4 * n, n, generator=g) / n ** 0.5
    bias = th.randn(4 * n, generator=g).to(dev)
    u3, h3 = image(padded(u.to(dev), p4(nin)), nin), image(padded(h.to(dev), p4(n)), n)
    wih3, whh3 = image(padded(wih.to(dev), p4(nin)), nin), image(padded(whh.to(dev), p4(n)), n)
    cpd = padded(cprev.to(dev), p4(n))
    for variant in (1, 2, 3, 4):
        hn, cn = th.zeros(m, p4(n), device=dev), th.zeros(m, p4(n), device=dev)
        gt = th.zeros(m, p4(4 * n), device=dev)
        h3n = th.zeros(lib.marl_image_bytes(m, n) + 256, dtype=th.uint8, device=dev)
        call = lambda: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(),
                                                  bias.data_ptr(), cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(),
                                                  gt.data_ptr(), h3n.data_ptr(), m, n, p4(n), p4(4 * n), variant, 2, None))
        us = timeit(call)
        rows.append(dict(kind="lstm_two_cells", m=m, n=n, nin=nin, variant=variant, us=round(us, 1),
                         tf=round(2 * 2.0 * m * 4 * n * (nin + n) / us / 1e6, 1)))
        print(rows[-1], flush=True)

for m in (512, 2048, 4096):
    k = 1024
    g = th.Generator().manual_seed(5 + m)
    As = [padded(th.randn(m, k, generator=g).to(dev), k) for _ in range(2)]
    Bs = [padded((th.randn(n, k, generator=g) / 32).to(dev), k) for n in (256, 256, 112, 112)]
    A3, B3 = [image(x, k) for x in As], [image(x, k) for x in Bs]
    Cs = [th.zeros(m, p4(n), device=dev) for n in (256, 256, 112, 112)]
    arr = lambda xs: (C.c_void_p * 4)(*[x.data_ptr() for x in xs])  # noqa: E731
    a3p, b3p, cp = arr([A3[0], A3[1], A3[0], A3[1]]), arr(B3), arr(Cs)
    ns = (C.c_int * 4)(256, 256, 112, 112)
    ldc = (C.c_int * 4)(*[c.shape[1] for c in Cs])
    ref = As[0][:256].cpu().double() @ Bs[2].cpu().double().t()
    for variant in (2, 3, 7, 10, 11, 12, 13):
        fn = lambda: check(lib.marl_gemm_nt_images_batch(4, a3p, b3p, cp, ns, ldc, m, k, 0, variant, None))
        us = timeit(fn)
        err = (Cs[2][:256, :112].cpu().double() - ref).abs().max().item()
        rows.append(dict(kind="nt_batch4", m=m, k=k, variant=variant, us=round(us, 1),
                         tf=round(2.0 * m * 736 * k / us / 1e6, 1), max_err=err))
        print(rows[-1], flush=True)

# batched heads / dU shapes at NR = 8192 (C4 at 32 images): [8192, 320, 256] x 2, [8192, 512, 1024]
for m, n, k in ((8192, 320, 256), (8192, 512, 1024), (65536, 45, 384), (65536, 384, 45), (65536, 384, 256), (65536, 256, 384), (65536, 256, 1024), (4096, 1024, 624)):
    g = th.Generator().manual_seed(m + n + k)
    a, b = th.randn(m, k, generator=g), th.randn(n, k, generator=g) / k ** 0.5
    a3, b3 = image(padded(a.to(dev), p4(k)), k), image(padded(b.to(dev), p4(k)), k)
    c1 = th.zeros(m, p4(n), device=dev)
    for variant in (1, 2, 3, 7, 10, 11, 12):
        fn = lambda: check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), None, c1.data_ptr(), c1.shape[1], m, n, k, 0, variant, None))
        us = timeit(fn)
        rows.append(dict(kind="nt", m=m, n=n, k=k, variant=variant, us=round(us, 1), tf=round(2.0 * m * n * k / us / 1e6, 1)))
        print(rows[-1], flush=True)

os.makedirs("gpurun_out", exist_ok=True)
json.dump(rows, open("gpurun_out/small_r_lab.json", "w"), indent=1)
