"""What bounds the 256 x 256 row-contraction image kernel: the same launch with the LDS-DMA, the MFMAs or the
fragment reads compiled out (library built with EXTRA=-DMARL_G3_ABLATE -> tools/bin/libmarl_abl.so).
python tools/tn_ablate.py"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlclassification_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libmarl_abl.so")
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa

out = []
names = {0: "product", 1: "no DMA", 2: "no MFMA", 3: "no fragment reads", 4: "no DMA, no fragment reads"}
for r, wgs in ((65536, 256), (8192, 32)):
    ni, nj = 1024, 256
    g = th.Generator().manual_seed(r)
    a3 = image(padded(th.randn(r, ni, generator=g).to(dev), ni), ni)
    b3 = image(padded(th.randn(r, nj, generator=g).to(dev), nj), nj)
    check(lib.marl_tune(b"g3_tn_variant", 3))
    check(lib.marl_tune(b"g3_tn_wgs", wgs))
    c1 = th.zeros(ni, nj, device=dev)
    sb3 = lib.marl_gemm_tn_images_scratch(ni, nj, r)
    sc3 = th.zeros(sb3 // 4 + 16, device=dev)
    for abl in (0, 1, 2, 3, 4, 0):
        check(lib.marl_tune(b"g3_tn_abl", abl))
        fn = lambda: check(lib.marl_gemm_tn_images(a3.data_ptr(), b3.data_ptr(), c1.data_ptr(), nj, ni, nj, r, None, sc3.data_ptr(), sb3, None))
        us = timeit(fn)
        out.append(dict(rows=r, workgroups=wgs, mode=names[abl], us=round(us, 1)))
        print(out[-1], flush=True)
json.dump(out, open("gpurun_out/tn_ablate.json", "w"), indent=1)
