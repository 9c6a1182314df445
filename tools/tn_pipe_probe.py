"""Round 6: does the phase-pipelined step keep the matrix pipe fed when memory is NOT the limit?  The 256 x 256
row-contraction launch of tools/tn_ablate.py on 32 workgroups (one CU in eight, 64 steps each: r5 measured 140 us for the
product against 98 us for MFMAs + barriers only) and on 256, pipe 0 / 1 interleaved.   python tools/tn_pipe_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th  # noqa: E402
from g3_lab import image, padded, timeit, lib, check, dev  # noqa: E402

out = []
for r, wgs in ((8192, 32), (65536, 256)):
    ni, nj = 1024, 256
    g = th.Generator().manual_seed(r)
    a3 = image(padded(th.randn(r, ni, generator=g).to(dev), ni), ni)
    b3 = image(padded(th.randn(r, nj, generator=g).to(dev), nj), nj)
    check(lib.marl_tune(b"g3_tn_variant", 3))
    check(lib.marl_tune(b"g3_tn_wgs", wgs))
    c1 = th.zeros(ni, nj, device=dev)
    sb3 = lib.marl_gemm_tn_images_scratch(ni, nj, r)
    sc3 = th.zeros(sb3 // 4 + 16, device=dev)
    fn = lambda: check(lib.marl_gemm_tn_images(a3.data_ptr(), b3.data_ptr(), c1.data_ptr(), nj, ni, nj, r, None, sc3.data_ptr(), sb3, None))
    for rep in range(3):
        for pipe in (0, 1):
            check(lib.marl_tune(b"g3_tn_pipe", pipe))
            out.append(dict(rows=r, workgroups=wgs, pipe=pipe, us=round(timeit(fn, 40), 1)))
            print(out[-1], flush=True)
json.dump(out, open("gpurun_out/tn_pipe_probe.json", "w"), indent=1)
