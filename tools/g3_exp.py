import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa
NAMES = {0: "product", 2: "no-dma", 3: "no-mfma", 4: "blockmajor", 6: "no-mfma+blockmajor"}
g = th.Generator().manual_seed(1)
m, n = 4096, 2048
for k in (320, 1280):
    ad = padded(th.randn(m, k, generator=g).to(dev), p4(k))
    bd = padded((th.randn(n, k, generator=g) / k ** 0.5).to(dev), p4(k))
    a3, b3 = image(ad, k), image(bd, k)
    c1 = th.zeros(m, p4(n), device=dev)
    for variant in (2, 4, 5, 1, 6):
        for abl in (0, 2, 3, 4, 6):
            check(lib.marl_tune(b"g3_safe", abl))
            fn = lambda: check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), None, c1.data_ptr(), c1.shape[1], m, n, k, 0, variant, None))
            us = timeit(fn)
            print(f"nt m={m} n={n} k={k} variant={variant} {NAMES[abl]:20s} {us:8.1f} us", flush=True)
            if k == 1280:
                check(lib.marl_tune(b"g3_clk", 1)); fn(); th.cuda.synchronize(); check(lib.marl_tune(b"g3_clk", 0))
check(lib.marl_tune(b"g3_safe", 0))
