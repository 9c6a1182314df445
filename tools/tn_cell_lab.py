"""Weight gradients of one LSTM cell: the four-launch form of round 4 (column passes over U + a 256 x 256 launch
over H) against the one-launch cell kernel, same operands.   python tools/tn_cell_lab.py"""
import os, sys, json
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa

out = []
for rows, ni, nih, nhh in ((65536, 1024, 368, 256), (65536, 1024, 624, 256)):
    gen = th.Generator().manual_seed(rows + nih)
    g3 = image(padded(th.randn(rows, ni, generator=gen).to(dev), ni), ni)
    u3 = image(padded(th.randn(rows, nih, generator=gen).to(dev), p4(nih)), nih)
    h3 = image(padded(th.randn(rows, nhh, generator=gen).to(dev), nhh), nhh)
    c_ih, c_hh, cs = th.zeros(ni, p4(nih), device=dev), th.zeros(ni, nhh, device=dev), th.zeros(ni, device=dev)
    sb1, sb2 = lib.marl_gemm_tn_images_scratch(ni, nih, rows), lib.marl_gemm_tn_images_scratch(ni, nhh, rows)
    s1, s2 = th.zeros(sb1 // 4 + 16, device=dev), th.zeros(sb2 // 4 + 16, device=dev)

    def old():
        check(lib.marl_gemm_tn_images(g3.data_ptr(), u3.data_ptr(), c_ih.data_ptr(), c_ih.shape[1], ni, nih, rows, None, s1.data_ptr(), sb1, None))
        check(lib.marl_gemm_tn_images(g3.data_ptr(), h3.data_ptr(), c_hh.data_ptr(), nhh, ni, nhh, rows, cs.data_ptr(), s2.data_ptr(), sb2, None))

    us_old = timeit(old, 20)
    ref_ih, ref_hh = c_ih.clone(), c_hh.clone()
    for splits, order in ((64, 0), (0, 2), (64, 0), (0, 2), (60, 2), (90, 2)):
        check(lib.marl_tune(b"g3_tn_cell_splits", splits))
        check(lib.marl_tune(b"g3_tn_cell_teams", 1 if order == 2 else 0))
        sb = lib.marl_gemm_tn_images_cell_scratch(ni, nih, nhh, rows)
        sc = th.zeros(sb // 4 + 16, device=dev)
        new = lambda: check(lib.marl_gemm_tn_images_cell(g3.data_ptr(), ni, u3.data_ptr(), nih, h3.data_ptr(), nhh, rows, c_ih.data_ptr(), c_ih.shape[1], c_hh.data_ptr(), nhh, cs.data_ptr(), sc.data_ptr(), sb, None))
        us = timeit(new, 20)
        d = max((c_ih - ref_ih).abs().max().item(), (c_hh - ref_hh).abs().max().item())
        out.append(dict(rows=rows, nih=nih, nhh=nhh, splits=splits, order=order, us_cell=round(us, 1), us_two_launches=round(us_old, 1), max_diff=d,
                        tf=round(2.0 * rows * ni * (nih + nhh) / us / 1e6, 1)))
        print(out[-1], flush=True)
json.dump(out, open("gpurun_out/tn_cell_lab.json", "w"), indent=1)
