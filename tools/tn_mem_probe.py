"""Is the row-contraction image kernel memory-bound?  Same tile plan and rows per split, operands that fit the
Infinity Cache (rows = 16384: A 100 MB) against the full 65536 rows (A 403 MB).  python tools/tn_mem_probe.py"""
import os, sys, json
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa

out = []
for nj, variant in ((256, 3), (368, 4), (368, 3), (624, 4)):
    for r in (8192, 16384, 32768, 65536):
        ni = 1024
        g = th.Generator().manual_seed(r + nj)
        a3 = image(padded(th.randn(r, ni, generator=g).to(dev), ni), ni)
        b3 = image(padded(th.randn(r, nj, generator=g).to(dev), p4(nj)), nj)
        check(lib.marl_tune(b"g3_tn_variant", variant))
        check(lib.marl_tune(b"g3_tn_wgs", 256 * r // 65536))  # same rows per split (1024) at every size
        c1 = th.zeros(ni, p4(nj), device=dev)
        sb3 = lib.marl_gemm_tn_images_scratch(ni, nj, r)
        sc3 = th.zeros(sb3 // 4 + 16, device=dev)
        fn = lambda: check(lib.marl_gemm_tn_images(a3.data_ptr(), b3.data_ptr(), c1.data_ptr(), c1.shape[1], ni, nj, r, None, sc3.data_ptr(), sb3, None))
        us = timeit(fn)
        out.append(dict(rows=r, nj=nj, variant=variant, us=round(us, 1), tf=round(2.0 * r * ni * nj / us / 1e6, 1), us_per_8k_rows=round(us * 8192 / r, 1)))
        print(out[-1], flush=True)
        del a3, b3
json.dump(out, open("gpurun_out/tn_mem_probe.json", "w"), indent=1)
