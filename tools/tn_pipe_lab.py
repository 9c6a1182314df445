"""Round 6: the row-contraction image kernels with (a) the LDS-DMA issued by inline asm (no compiler vmcnt(0) in front
of the transposing reads) and (b) the phase-pipelined step (knob g3_tn_pipe), against the round-5 library
(tools/bin/libmarl_r5.so, run as a second process: MARL_LAB_LIB=...).  Checks: pipelined == fully-waited build, bit
for bit; pipe 1 == pipe 0.        python tools/tn_pipe_lab.py            (on the GPU box)"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from marlclassification_amd import _lib  # noqa: E402

if os.environ.get("MARL_LAB_LIB"):
    _lib.LIB_PATH = os.environ["MARL_LAB_LIB"]
import torch as th  # noqa: E402

sys.path.insert(0, HERE)
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa: E402

old = bool(os.environ.get("MARL_LAB_LIB"))
out = []


def tune(k, v):
    check(lib.marl_tune(k.encode(), v))


for rows, ni, nih, nhh in ((65536, 1024, 368, 256), (65536, 1024, 624, 256), (8192, 1024, 624, 256)):
    gen = th.Generator().manual_seed(rows + nih)
    g3 = image(padded(th.randn(rows, ni, generator=gen).to(dev), ni), ni)
    u3 = image(padded(th.randn(rows, nih, generator=gen).to(dev), p4(nih)), nih)
    h3 = image(padded(th.randn(rows, nhh, generator=gen).to(dev), nhh), nhh)
    c_ih, c_hh, cs = th.zeros(ni, p4(nih), device=dev), th.zeros(ni, nhh, device=dev), th.zeros(ni, device=dev)
    res = {}
    for pipe in ((0,) if old else (0, 1, 0, 1)):
        if not old:
            tune("g3_tn_pipe", pipe)
        if rows >= 8192 and lib.marl_gemm_tn_images_cell_scratch(ni, nih, nhh, rows) > 0:
            sb = lib.marl_gemm_tn_images_cell_scratch(ni, nih, nhh, rows)
            sc = th.zeros(sb // 4 + 16, device=dev)
            cell = lambda: check(lib.marl_gemm_tn_images_cell(g3.data_ptr(), ni, u3.data_ptr(), nih, h3.data_ptr(), nhh, rows, c_ih.data_ptr(),
                                                              c_ih.shape[1], c_hh.data_ptr(), nhh, cs.data_ptr(), sc.data_ptr(), sb, None))
            tune("g3_safe", 1)
            cell()
            th.cuda.synchronize()
            safe = (c_ih.clone(), c_hh.clone(), cs.clone())
            tune("g3_safe", 0)
            same = True
            for _ in range(3):
                c_ih.zero_(), c_hh.zero_(), cs.zero_()
                cell()
                th.cuda.synchronize()
                same = same and all(th.equal(a, b) for a, b in zip((c_ih, c_hh, cs), safe))
            us = timeit(cell, 20)
            key = "cell"
            eq0 = all(th.equal(a, b) for a, b in zip(safe, res[key])) if key in res else None
            res.setdefault(key, safe)
            out.append(dict(kind="cell", rows=rows, ni=ni, nih=nih, nhh=nhh, pipe=pipe, lib="r5" if old else "r6", us=round(us, 1),
                            tf=round(2.0 * rows * ni * (nih + nhh) / us / 1e6, 1), pipelined_eq_safe=same, eq_first_mode=eq0))
            print(out[-1], flush=True)
        for nj, b3 in ((nhh, h3), (nih, u3)):  # one product: 256 x 256 tiles (nj = 256) / column passes (368, 624)
            c1 = th.zeros(ni, p4(nj), device=dev)
            cs1 = th.zeros(ni, device=dev)
            sb3 = lib.marl_gemm_tn_images_scratch(ni, nj, rows)
            sc3 = th.zeros(sb3 // 4 + 16, device=dev)
            one = lambda: check(lib.marl_gemm_tn_images(g3.data_ptr(), b3.data_ptr(), c1.data_ptr(), c1.shape[1], ni, nj, rows, cs1.data_ptr(),
                                                        sc3.data_ptr(), sb3, None))
            tune("g3_safe", 1)
            one()
            th.cuda.synchronize()
            safe = (c1.clone(), cs1.clone())
            tune("g3_safe", 0)
            same = True
            for _ in range(3):
                c1.zero_(), cs1.zero_()
                one()
                th.cuda.synchronize()
                same = same and th.equal(c1, safe[0]) and th.equal(cs1, safe[1])
            us = timeit(one, 20)
            key = ("one", nj)
            eq0 = (th.equal(safe[0], res[key][0]) and th.equal(safe[1], res[key][1])) if key in res else None
            res.setdefault(key, safe)
            out.append(dict(kind="tn", rows=rows, ni=ni, nj=nj, pipe=pipe, lib="r5" if old else "r6", us=round(us, 1),
                            tf=round(2.0 * rows * ni * nj / us / 1e6, 1), pipelined_eq_safe=same, eq_first_mode=eq0))
            print(out[-1], flush=True)
if not old:
    r5 = os.path.join(HERE, "bin", "libmarl_r5.so")
    if os.path.exists(r5):
        p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, MARL_LAB_LIB=r5), capture_output=True, text=True)
        print(p.stdout[-3000:], p.stderr[-500:])
        try:
            out += json.load(open("gpurun_out/tn_pipe_lab_r5.json"))
        except Exception as e:  # noqa
            print("no r5 record", e)
json.dump(out, open("gpurun_out/tn_pipe_lab_r5.json" if old else "gpurun_out/tn_pipe_lab.json", "w"), indent=1)
