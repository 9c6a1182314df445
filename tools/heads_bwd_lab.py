"""Round 6: would the backward head products pay on operand images?  Today (fp32 operands split while staged): dH / dH^
= dZ W0 over the three heads in one launch (275 us at C3) and three weight gradients dZ^T H (107 us each).  Here the
same shapes on the image kernels: NT [65536 x 256 x 384] x 3 (phase-pipelined 256 x 256 plan, accumulate) and TN
[65536 rows; 384 x 256] x 3.    python tools/heads_bwd_lab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th
from g3_lab import image, padded, timeit, p4, lib, check, dev
rows, nl, nh = 65536, 384, 256
g = th.Generator().manual_seed(3)
dz = th.randn(rows, nl, generator=g).to(dev)
h = th.randn(rows, nh, generator=g).to(dev)
w0t = (th.randn(nh, nl, generator=g) / 20).to(dev)  # W0^T: [n_in, n_hidden] -> B operand of dH = dZ W0
dz3, h3, w3 = image(dz, nl), image(h, nh), image(w0t, nl)
out = th.zeros(rows, nh, device=dev)
for variant in (2, 21, 22):
    fn = lambda: check(lib.marl_gemm_nt_images(dz3.data_ptr(), w3.data_ptr(), None, out.data_ptr(), nh, rows, nh, nl, 1, variant, None))
    print(f"NT  dH += dZ W0  [65536 x 256 x 384] variant {variant}: {timeit(fn, 20):6.1f} us", flush=True)
c1, cs = th.zeros(nl, nh, device=dev), th.zeros(nl, device=dev)
sb = lib.marl_gemm_tn_images_scratch(nl, nh, rows)
sc = th.zeros(sb // 4 + 16, device=dev)
for tv in (0, 2, 3):
    check(lib.marl_tune(b"g3_tn_variant", tv))
    sb = lib.marl_gemm_tn_images_scratch(nl, nh, rows)
    sc = th.zeros(sb // 4 + 16, device=dev)
    fn = lambda: check(lib.marl_gemm_tn_images(dz3.data_ptr(), h3.data_ptr(), c1.data_ptr(), nh, nl, nh, rows, cs.data_ptr(), sc.data_ptr(), sb, None))
    print(f"TN  dW0 = dZ^T H [65536 rows; 384 x 256] tn variant {tv}: {timeit(fn, 20):6.1f} us (incl. slab reduce)", flush=True)
check(lib.marl_tune(b"g3_tn_variant", 0))
# swapped roles: H^T dZ (256 x 384: column passes 256 + 128) - the transposed result
sb = lib.marl_gemm_tn_images_scratch(nh, nl, rows)
sc = th.zeros(sb // 4 + 16, device=dev)
c2 = th.zeros(nh, nl, device=dev)
fn = lambda: check(lib.marl_gemm_tn_images(h3.data_ptr(), dz3.data_ptr(), c2.data_ptr(), nl, nh, nl, rows, None, sc.data_ptr(), sb, None))
print(f"TN  (H^T dZ, 256 x 384, column passes): {timeit(fn, 20):6.1f} us", flush=True)
