import os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa
r, ni, nj = 65536, 1024, 368
g = th.Generator().manual_seed(1)
ad, bd = padded(th.randn(r, ni, generator=g).to(dev), p4(ni)), padded(th.randn(r, nj, generator=g).to(dev), p4(nj))
a3, b3 = image(ad, ni), image(bd, nj)
c1 = th.zeros(ni, p4(nj), device=dev)
sb3 = lib.marl_gemm_tn_images_scratch(ni, nj, r)
sc3 = th.zeros(sb3 // 4 + 16, device=dev)
for _ in range(3):
    check(lib.marl_gemm_tn_images(a3.data_ptr(), b3.data_ptr(), c1.data_ptr(), c1.shape[1], ni, nj, r, None, sc3.data_ptr(), sb3, None))
# NT for comparison (same tile structure)
m, n, k = 4096, 2048, 1280
ad, bd = padded(th.randn(m, k, generator=g).to(dev), p4(k)), padded(th.randn(n, k, generator=g).to(dev), p4(k))
a3, b3 = image(ad, k), image(bd, k)
c1 = th.zeros(m, p4(n), device=dev)
for _ in range(3):
    check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), None, c1.data_ptr(), c1.shape[1], m, n, k, 0, 2, None))
th.cuda.synchronize()
