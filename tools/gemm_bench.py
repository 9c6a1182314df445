"""Micro-benchmark of the NT GEMM through the C ABI (perf debugging aid).
usage: python tools/gemm_bench.py M N K [iters]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch as th  # noqa: E402

from marlclassification_amd import _lib  # noqa: E402

m, n, k = (int(v) for v in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib = _lib.load()
dev = th.device("cuda:0")
a = th.randn(m, k, device=dev)
b = th.randn(n, k, device=dev)
c = th.zeros(m, n, device=dev)
for _ in range(3):
    lib.marl_gemm_nt(a.data_ptr(), k, b.data_ptr(), k, None, c.data_ptr(), n, m, n, k, 0, None)
th.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    lib.marl_gemm_nt(a.data_ptr(), k, b.data_ptr(), k, None, c.data_ptr(), n, m, n, k, 0, None)
th.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"M={m} N={n} K={k}: {dt * 1e6:.1f} us, {2.0 * m * n * k / dt / 1e12:.1f} TFLOP/s")
ref = (a[:64].double() @ b.double().t()).float()
print("max err", (c[:64] - ref).abs().max().item())
