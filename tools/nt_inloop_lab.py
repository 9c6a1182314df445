"""Round 6: the in-loop backward batch (4 products, M = 4096, K = 1024, n = 256 / 256 / 112 / 112) on the tile plans:
7 = 64 x 64 (shipped), 3 = 128 x 64, 2 = 128 x 128; 23 / 24 = phase-pipelined 128 x 64 / 128 x 128 (round-6 experiment: needs the two
launch_g3p_variant lines of commit "memory-side counters ..." in launch_gemm_nt3; measured 41.2 / 40.6 us against 39.0 for 64 x 64)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th
from g3_lab import image, padded, timeit, p4, lib, check, dev
m, k = 4096, 1024
g = th.Generator().manual_seed(5)
As = [padded(th.randn(m, k, generator=g).to(dev), k) for _ in range(2)]
Bs = [padded((th.randn(n, k, generator=g) / 32).to(dev), k) for n in (256, 256, 112, 112)]
A3 = [image(x, k) for x in As]; B3 = [image(x, k) for x in Bs]
Cs = [th.zeros(m, p4(n), device=dev) for n in (256, 256, 112, 112)]
arr = lambda xs: (C.c_void_p * 4)(*[x.data_ptr() for x in xs])
a3p, b3p, cp = arr([A3[0], A3[1], A3[0], A3[1]]), arr(B3), arr(Cs)
ns = (C.c_int * 4)(256, 256, 112, 112); ldc = (C.c_int * 4)(*[c.shape[1] for c in Cs])
ref = None
for rep in range(3):
    for variant in (7, 3, 2, 23, 24):
        fn = lambda: check(lib.marl_gemm_nt_images_batch(4, a3p, b3p, cp, ns, ldc, m, k, 0, variant, None))
        fn(); th.cuda.synchronize()
        cur = [c.clone() for c in Cs]
        if ref is None: ref = cur
        same = all(th.equal(a, b) for a, b in zip(ref, cur))
        print(f"variant {variant:2d}: {timeit(fn, 50):6.1f} us  bit-equal to 64x64: {same}", flush=True)
