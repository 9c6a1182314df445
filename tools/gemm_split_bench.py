"""Times the matrix kernels of libmarl_hip.so in both forms (bf16x6 split / exact-fp32 MFMA) on
the product shapes of the C3 iteration and reports their error against float64.
    python tools/gemm_split_bench.py            (on the GPU box)"""
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


def padded(t, ld):
    out = th.zeros(t.shape[0], ld, device=dev)
    out[:, : t.shape[1]] = t
    return out


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


rows = []
NT = [(65536, 384, 256), (65536, 256, 384), (65536, 256, 1024), (4096, 256, 1024), (4096, 112, 1024),
      (65536, 45, 384), (65536, 384, 45), (4096, 1024, 624)]
for m, n, k in NT:
    g = th.Generator().manual_seed(m + n + k)
    a = th.randn(m, k, generator=g)
    b = th.randn(n, k, generator=g) / k ** 0.5
    ws = th.zeros(m * p4(k) + 64, device=dev)
    ad = ws[: m * p4(k)].view(m, p4(k))
    ad[:, :k] = a.to(dev)
    bd = padded(b.to(dev), p4(k))
    ref = (a[:2048].double() @ b.double().t())
    for mode in (0, 1):
        check(lib.marl_tune(b"mfma_split", mode))
        cd = th.zeros(m, p4(n), device=dev)
        img = th.zeros(lib.marl_gemm_weight_image_bytes(n, k) // 4 + 64, device=dev)
        fn = lambda: check(lib.marl_gemm_nt_weights(ad.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1], None,
                                                    cd.data_ptr(), cd.shape[1], m, n, k, 0, img.data_ptr(), None))
        us = timeit(fn)
        err = (cd[:2048, :n].cpu().double() - ref).abs().max().item()
        rows.append(dict(kind="nt", m=m, n=n, k=k, split=mode, us=round(us, 1),
                         tflops=round(2.0 * m * n * k / us / 1e6, 1), max_err=err))
        print(rows[-1], flush=True)
TN = [(65536, 1024, 368), (65536, 1024, 256), (65536, 384, 256), (65536, 128, 256)]
for r, ni, nj in TN:
    g = th.Generator().manual_seed(r + ni + nj)
    a = th.randn(r, ni, generator=g)
    b = th.randn(r, nj, generator=g)
    ad, bd = padded(a.to(dev), p4(ni)), padded(b.to(dev), p4(nj))
    ref = a.double().t() @ b.double()
    for mode in (0, 1):
        check(lib.marl_tune(b"mfma_split", mode))
        cd = th.zeros(ni, p4(nj), device=dev)
        sb = lib.marl_gemm_tn_scratch(ni, nj, r)
        scratch = th.zeros(sb // 4 + 16, device=dev)
        fn = lambda: check(lib.marl_gemm_tn(ad.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1],
                                            cd.data_ptr(), cd.shape[1], ni, nj, r, scratch.data_ptr(), sb, None))
        us = timeit(fn)
        err = (cd[:, :nj].cpu().double() - ref).abs().max().item()
        rows.append(dict(kind="tn", rows=r, ni=ni, nj=nj, split=mode, us=round(us, 1),
                         tflops=round(2.0 * r * ni * nj / us / 1e6, 1), max_err=err,
                         ref_max=ref.abs().max().item()))
        print(rows[-1], flush=True)
check(lib.marl_tune(b"mfma_split", 1))
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/gemm_split_bench.json", "w") as f:
    json.dump(rows, f, indent=1)
