"""Ablation timings of the image GEMM kernels (library built with EXTRA=-DMARL_G3_ABLATE)."""
import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlclassification_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libmarl_abl.so")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa

NAMES = {0: "product", 1: "safe", 2: "no-dma", 3: "no-mfma", 4: "blockmajor", 5: "no-dsread"}
for (m, n, k, variants) in ((65536, 256, 1024, (1, 2)), (4096, 1024, 624, (1, 2, 3))):
    g = th.Generator().manual_seed(1)
    ad = padded(th.randn(m, k, generator=g).to(dev), p4(k))
    bd = padded((th.randn(n, k, generator=g) / k ** 0.5).to(dev), p4(k))
    a3, b3 = image(ad, k), image(bd, k)
    c1 = th.zeros(m, p4(n), device=dev)
    for variant in variants:
        for abl in (0, 1, 2, 3, 4, 5):
            check(lib.marl_tune(b"g3_safe", abl))
            us = timeit(lambda: check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), None, c1.data_ptr(), c1.shape[1], m, n, k, 0, variant, None)))
            print(f"nt m={m} n={n} k={k} variant={variant} {NAMES[abl]:10s} {us:8.1f} us", flush=True)
check(lib.marl_tune(b"g3_safe", 0))
m, n, nin = 4096, 256, 368
g = th.Generator().manual_seed(2)
u3 = image(padded(th.randn(m, nin, generator=g).to(dev), p4(nin)), nin)
h3 = image(padded(th.randn(m, n, generator=g).to(dev), p4(n)), n)
wih3 = image(padded((th.randn(4 * n, nin, generator=g) / 19).to(dev), p4(nin)), nin)
whh3 = image(padded((th.randn(4 * n, n, generator=g) / 16).to(dev), p4(n)), n)
cpd, bd = th.randn(m, n, generator=g).to(dev), th.randn(4 * n, generator=g).to(dev)
hn, cn, gt = th.zeros(m, n, device=dev), th.zeros(m, n, device=dev), th.zeros(m, 4 * n, device=dev)
h3n = th.zeros(lib.marl_image_bytes(m, n) + 256, dtype=th.uint8, device=dev)
for variant in (1, 2):
    for abl in (0, 1, 2, 3, 4, 5):
        check(lib.marl_tune(b"g3_safe", abl))
        for cells in (1, 2):
            us = timeit(lambda: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(), bd.data_ptr(),
                                                           cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(), gt.data_ptr(), h3n.data_ptr(), m, n, n, 4 * n,
                                                           variant, cells, None)))
            print(f"lstm variant={variant} cells={cells} {NAMES[abl]:10s} {us:8.1f} us", flush=True)
check(lib.marl_tune(b"g3_safe", 0))
