"""Launch time of the image GEMMs against the K depth: slope = cost of a 16-deep step, intercept = launch + prologue + epilogue."""
import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa

g = th.Generator().manual_seed(1)
for (m, n, variant) in ((4096, 1024, 2), (4096, 2048, 2), (4096, 2048, 1), (65536, 256, 1)):
    for k in (64, 320, 640, 1280):
        ad = padded(th.randn(m, k, generator=g).to(dev), p4(k))
        bd = padded((th.randn(n, k, generator=g) / k ** 0.5).to(dev), p4(k))
        a3, b3 = image(ad, k), image(bd, k)
        c1 = th.zeros(m, p4(n), device=dev)
        us = timeit(lambda: check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), None, c1.data_ptr(), c1.shape[1], m, n, k, 0, variant, None)))
        print(f"nt m={m} n={n} variant={variant} k={k:5d} steps={k//16:3d} {us:8.1f} us", flush=True)
m, n = 4096, 256
for variant in (2, 1):
    for nin in (64, 368, 1024):
        u3 = image(padded(th.randn(m, nin, generator=g).to(dev), p4(nin)), nin)
        h3 = image(padded(th.randn(m, n, generator=g).to(dev), p4(n)), n)
        wih3 = image(padded((th.randn(4 * n, nin, generator=g) / 19).to(dev), p4(nin)), nin)
        whh3 = image(padded((th.randn(4 * n, n, generator=g) / 16).to(dev), p4(n)), n)
        cpd, bd = th.randn(m, n, generator=g).to(dev), th.randn(4 * n, generator=g).to(dev)
        hn, cn, gt = th.zeros(m, n, device=dev), th.zeros(m, n, device=dev), th.zeros(m, 4 * n, device=dev)
        h3n = th.zeros(lib.marl_image_bytes(m, n) + 256, dtype=th.uint8, device=dev)
        for gates in (1, 0):
            for cells in (1, 2):
                us = timeit(lambda: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(), bd.data_ptr(),
                                                               cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(), gt.data_ptr() if gates else None,
                                                               h3n.data_ptr() if gates else None, m, n, n, 4 * n, variant, cells, None)))
                print(f"lstm variant={variant} nin={nin:5d} steps={(nin + n) // 16:3d} cells={cells} gates+image={gates} {us:8.1f} us", flush=True)
