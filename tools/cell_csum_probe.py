"""Round 6: does the column-sum workgroup (6 extra MFMAs per step on the first 256-wide tile of every G tile) pull its
team out of step?  The cell weight-gradient launch with and without the column sums asked for.  python tools/cell_csum_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th
from g3_lab import image, padded, timeit, p4, lib, check, dev
rows, ni, nih, nhh = 65536, 1024, 368, 256
gen = th.Generator().manual_seed(5)
g3 = image(padded((th.randn(rows, ni, generator=gen) * 0.01).to(dev), ni), ni)
u3 = image(padded(th.randn(rows, nih, generator=gen).to(dev), p4(nih)), nih)
h3 = image(padded(th.randn(rows, nhh, generator=gen).to(dev), nhh), nhh)
c_ih, c_hh, cs = th.zeros(ni, p4(nih), device=dev), th.zeros(ni, nhh, device=dev), th.zeros(ni, device=dev)
sb = lib.marl_gemm_tn_images_cell_scratch(ni, nih, nhh, rows)
sc = th.zeros(sb // 4 + 16, device=dev)
for rep in range(3):
    for name, csp in (("with column sums", cs.data_ptr()), ("without", None)):
        fn = lambda: check(lib.marl_gemm_tn_images_cell(g3.data_ptr(), ni, u3.data_ptr(), nih, h3.data_ptr(), nhh, rows, c_ih.data_ptr(), c_ih.shape[1], c_hh.data_ptr(), nhh, csp, sc.data_ptr(), sb, None))
        print(f"{name:18s} {timeit(fn, 30):7.1f} us", flush=True)
