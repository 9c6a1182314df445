"""Runs one product of libmarl_hip.so a few times (profiling / timestamp runs).
usage: python tools/ts_run.py nt M N K | tn ROWS NI NJ"""
import os, sys, torch as th
sys.path.insert(0, os.getcwd())
from marlclassification_amd import _lib
lib, check = _lib.load(), _lib.check
dev = th.device("cuda:0")
kind = sys.argv[1]
x, y, z = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if kind == "nt":
    m, n, k = x, y, z
    a = th.randn(m, k, device=dev); b = th.randn(n, k, device=dev); c = th.zeros(m, n, device=dev)
    img = th.zeros(lib.marl_gemm_weight_image_bytes(n, k) // 4 + 64, device=dev)
    for i in range(8):
        check(lib.marl_gemm_nt_weights(a.data_ptr(), k, b.data_ptr(), k, None, c.data_ptr(), n, m, n, k, 0, img.data_ptr(), None))
else:
    r, ni, nj = x, y, z
    a = th.randn(r, ni, device=dev); b = th.randn(r, nj, device=dev); c = th.zeros(ni, nj, device=dev)
    sb = lib.marl_gemm_tn_scratch(ni, nj, r)
    s = th.zeros(sb // 4 + 16, device=dev)
    for i in range(8):
        check(lib.marl_gemm_tn(a.data_ptr(), ni, b.data_ptr(), nj, c.data_ptr(), nj, ni, nj, r, s.data_ptr(), sb, None))
th.cuda.synchronize()
