"""time(K) of the NT product at fixed M, N: intercept = per-workgroup fixed cost, slope = cost per 32-deep tile."""
import os, sys, torch as th
sys.path.insert(0, os.getcwd())
from marlclassification_amd import _lib
lib, check = _lib.load(), _lib.check
dev = th.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(3): fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); th.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode in (1, 0):
    check(lib.marl_tune(b"mfma_split", mode))
    for (m, n) in [(4096, 1024), (32768, 256)]:
        res = []
        for k in (32, 64, 128, 320, 640, 1280):
            ws = th.randn(m * k + 64, device=dev); a = ws[: m * k].view(m, k)
            b = th.randn(n, k, device=dev); c = th.zeros(m, n, device=dev)
            img = th.zeros(lib.marl_gemm_weight_image_bytes(n, k) // 4 + 64, device=dev)
            fn = lambda: check(lib.marl_gemm_nt_weights(a.data_ptr(), k, b.data_ptr(), k, None, c.data_ptr(), n, m, n, k, 0, img.data_ptr(), None))
            res.append((k // 32, round(timeit(fn), 1)))
        slope = (res[-1][1] - res[-2][1]) / (res[-1][0] - res[-2][0])
        print("split" if mode else "fp32 ", (m, n), "tiles->us", res, "slope %.2f us/tile  intercept %.1f us" % (slope, res[-1][1] - slope * res[-1][0]), flush=True)
check(lib.marl_tune(b"mfma_split", 1))
