"""Image GEMMs (csrc/gemm3.hip) against the bf16x6 kernels of round 3 on the product shapes of the C3
iteration: bit-equality of the pipelined kernel with its fully-waited SAFE build, error against float64,
time per launch.      python tools/g3_lab.py [quick]      (on the GPU box)"""
import ctypes as C
import json
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlclassification_amd import _lib  # noqa: E402

lib, check = _lib.load(), _lib.check
dev = th.device("cuda:0")
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"


def p4(x):
    return (x + 3) & ~3


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


def image(t, k):
    """fp32 device matrix [rows, ld >= k] -> its k16 image (uint8 tensor)"""
    rows = t.shape[0]
    img = th.zeros(lib.marl_image_bytes(rows, k) + 256, dtype=th.uint8, device=dev)
    check(lib.marl_image_build(t.data_ptr(), t.shape[1], rows, k, img.data_ptr(), None))
    return img


def padded(t, ld):
    out = th.zeros(t.shape[0], ld, device=dev)
    out[:, : t.shape[1]] = t
    return out


def main():
    rows = []
    NT = [(4096, 256, 1024), (4096, 112, 1024), (65536, 384, 256), (65536, 256, 384), (65536, 256, 1024),
          (65536, 45, 384), (65536, 384, 45), (4096, 1024, 624), (1000, 70, 23)]
    if quick:
        NT = NT[:3] + NT[-1:]
    for m, n, k in NT:
        g = th.Generator().manual_seed(m + n + k)
        a = th.randn(m, k, generator=g)
        b = th.randn(n, k, generator=g) / k ** 0.5
        bias = th.randn(n, generator=g).to(dev)
        ad, bd = padded(a.to(dev), p4(k)), padded(b.to(dev), p4(k))
        a3, b3 = image(ad, k), image(bd, k)
        mr = min(m, 2048)
        ref = a[:mr].double() @ b.double().t() + bias.cpu().double()
        # round-3 kernel (fp32 A split while staged, weight image for B)
        cd = th.zeros(m, p4(n), device=dev)
        wimg = th.zeros(lib.marl_gemm_weight_image_bytes(n, k) // 4 + 64, device=dev)
        fn = lambda: check(lib.marl_gemm_nt_weights(ad.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1], bias.data_ptr(),
                                                    cd.data_ptr(), cd.shape[1], m, n, k, 0, wimg.data_ptr(), None))
        us0 = timeit(fn)
        err0 = (cd[:mr, :n].cpu().double() - ref).abs().max().item()
        for variant in (1, 2, 3):
            if variant in (1, 2) and n < 96 and m < 60000:
                pass
            c1 = th.full((m, p4(n)), 7.0, device=dev)
            fn = lambda: check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), bias.data_ptr(), c1.data_ptr(),
                                                       c1.shape[1], m, n, k, 0, variant, None))
            check(lib.marl_tune(b"g3_safe", 1))
            fn()
            th.cuda.synchronize()
            c_safe = c1.clone()
            check(lib.marl_tune(b"g3_safe", 0))
            same = True
            for _ in range(3):
                c1.fill_(7.0)
                fn()
                th.cuda.synchronize()
                same = same and bool(th.equal(c1[:, :n], c_safe[:, :n]))
            untouched = bool((c1[:, n:] == 7.0).all().item()) if p4(n) > n else True
            us = timeit(fn)
            err = (c1[:mr, :n].cpu().double() - ref).abs().max().item()
            rows.append(dict(kind="nt", m=m, n=n, k=k, variant=variant, us=round(us, 1), us_r3=round(us0, 1),
                             tf=round(2.0 * m * n * k / us / 1e6, 1), pipelined_eq_safe=same, pad_untouched=untouched,
                             max_err=err, max_err_r3=err0))
            print(rows[-1], flush=True)

    # the in-loop backward batch: 2 x [4096, 256, 1024] + 2 x [4096, 112, 1024] in one launch
    m, k = 4096, 1024
    g = th.Generator().manual_seed(5)
    As = [padded(th.randn(m, k, generator=g).to(dev), k) for _ in range(2)]
    Bs = [padded((th.randn(n, k, generator=g) / 32).to(dev), k) for n in (256, 256, 112, 112)]
    A3 = [image(x, k) for x in As]
    B3 = [image(x, k) for x in Bs]
    Cs = [th.zeros(m, p4(n), device=dev) for n in (256, 256, 112, 112)]
    arr = lambda xs: (C.c_void_p * 4)(*[x.data_ptr() for x in xs])
    a3p, b3p, cp = arr([A3[0], A3[1], A3[0], A3[1]]), arr(B3), arr(Cs)
    ns = (C.c_int * 4)(256, 256, 112, 112)
    ldc = (C.c_int * 4)(*[c.shape[1] for c in Cs])
    for variant in (1, 2, 3):
        fn = lambda: check(lib.marl_gemm_nt_images_batch(4, a3p, b3p, cp, ns, ldc, m, k, 0, variant, None))
        us = timeit(fn)
        ref = As[0][:512].cpu().double() @ Bs[2].cpu().double().t()
        err = (Cs[2][:512, :112].cpu().double() - ref).abs().max().item()
        rows.append(dict(kind="nt_batch4", variant=variant, us=round(us, 1), tf=round(2.0 * m * 736 * k / us / 1e6, 1), max_err=err))
        print(rows[-1], flush=True)

    # LSTM cell: m = 4096 rows, n = 256 units, nin = 368 (C3), two cells per launch for the timing
    for (m, n, nin) in ((4096, 256, 368), (777, 23, 45)):
        g = th.Generator().manual_seed(m + n)
        u, h, cprev = th.randn(m, nin, generator=g), th.randn(m, n, generator=g), th.randn(m, n, generator=g)
        wih, whh = th.randn(4 * n, nin, generator=g) / nin ** 0.5, th.randn(4 * n, n, generator=g) / n ** 0.5
        bias = th.randn(4 * n, generator=g)
        gates_ref = u.double() @ wih.double().t() + h.double() @ whh.double().t() + bias.double()
        i_, f_, g_, o_ = gates_ref.chunk(4, dim=1)
        c_ref = th.sigmoid(f_) * cprev.double() + th.sigmoid(i_) * th.tanh(g_)
        h_ref = th.sigmoid(o_) * th.tanh(c_ref)
        ud, hd = padded(u.to(dev), p4(nin)), padded(h.to(dev), p4(n))
        u3, h3 = image(ud, nin), image(hd, n)
        wih3, whh3 = image(padded(wih.to(dev), p4(nin)), nin), image(padded(whh.to(dev), p4(n)), n)
        cpd, bd = padded(cprev.to(dev), p4(n)), bias.to(dev)
        for variant in (1, 2):
            hn, cn = th.zeros(m, p4(n), device=dev), th.zeros(m, p4(n), device=dev)
            gt = th.zeros(m, p4(4 * n), device=dev)
            h3n = th.zeros(lib.marl_image_bytes(m, n) + 256, dtype=th.uint8, device=dev)
            call = lambda cells: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(),
                                                            bd.data_ptr(), cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(), gt.data_ptr(),
                                                            h3n.data_ptr(), m, n, p4(n), p4(4 * n), variant, cells, None))
            call(1)
            th.cuda.synchronize()
            eh = (hn[:, :n].cpu().double() - h_ref).abs().max().item()
            ec = (cn[:, :n].cpu().double() - c_ref).abs().max().item()
            want3 = image(hn, n)  # the image of the h' the kernel wrote must be what it wrote as an image
            nb = lib.marl_image_bytes(m, n)
            img_ok = bool(th.equal(want3[:nb], h3n[:nb]))
            us2 = timeit(lambda: call(2))
            us1 = timeit(lambda: call(1))
            rows.append(dict(kind="lstm", m=m, n=n, nin=nin, variant=variant, us_two_cells=round(us2, 1), us_one_cell=round(us1, 1),
                             tf_two=round(2 * 2.0 * m * 4 * n * (nin + n) / us2 / 1e6, 1), err_h=eh, err_c=ec, h_image_ok=img_ok))
            print(rows[-1], flush=True)

    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/g3_lab.json", "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
