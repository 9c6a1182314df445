"""TN image kernel (csrc/gemm3.hip) against the round-3 fp32-operand kernel: error vs float64, column sums, time."""
import os, sys, json
import torch as th
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa

rows_out = []
import sys as _s
SHAPES = ((65536, 1024, 624), (8192, 1024, 624), (8192, 1024, 256), (49152, 256, 96)) if len(_s.argv) > 1 else ((65536, 1024, 368), (65536, 1024, 256), (65536, 384, 256), (65536, 128, 256), (2048, 96, 80), (4096, 45, 384))
for r, ni, nj in SHAPES:
    g = th.Generator().manual_seed(r + ni + nj)
    a = th.randn(r, ni, generator=g)
    b = th.randn(r, nj, generator=g)
    ad, bd = padded(a.to(dev), p4(ni)), padded(b.to(dev), p4(nj))
    ref = a.double().t() @ b.double()
    cs_ref = a.double().sum(0)
    # round 3
    cd = th.zeros(ni, p4(nj), device=dev)
    sb = lib.marl_gemm_tn_scratch(ni, nj, r)
    scratch = th.zeros(sb // 4 + 16, device=dev)
    us0 = timeit(lambda: check(lib.marl_gemm_tn(ad.data_ptr(), ad.shape[1], bd.data_ptr(), bd.shape[1], cd.data_ptr(), cd.shape[1], ni, nj, r, scratch.data_ptr(), sb, None)))
    err0 = (cd[:, :nj].cpu().double() - ref).abs().max().item()
    a3, b3 = image(ad, ni), image(bd, nj)
    for variant in (2, 3, 4):
        check(lib.marl_tune(b"g3_tn_variant", variant))
        c1 = th.full((ni, p4(nj)), 7.0, device=dev)
        cs = th.zeros(ni, device=dev)
        sb3 = lib.marl_gemm_tn_images_scratch(ni, nj, r)
        sc3 = th.zeros(sb3 // 4 + 16, device=dev)
        fn = lambda: check(lib.marl_gemm_tn_images(a3.data_ptr(), b3.data_ptr(), c1.data_ptr(), c1.shape[1], ni, nj, r, cs.data_ptr(), sc3.data_ptr(), sb3, None))
        check(lib.marl_tune(b"g3_safe", 1)); fn(); th.cuda.synchronize(); c_safe = c1.clone(); check(lib.marl_tune(b"g3_safe", 0))
        same = True
        for _ in range(3):
            c1.fill_(7.0); fn(); th.cuda.synchronize()
            same = same and bool(th.equal(c1[:, :nj], c_safe[:, :nj]))
        us = timeit(fn)
        err = (c1[:, :nj].cpu().double() - ref).abs().max().item()
        ecs = (cs.cpu().double() - cs_ref).abs().max().item()
        row = dict(rows=r, ni=ni, nj=nj, variant=variant, us=round(us, 1), us_r3=round(us0, 1), tf=round(2.0 * r * ni * nj / us / 1e6, 1),
                   eq_safe=same, max_err=err, max_err_r3=err0, ref_max=ref.abs().max().item(), colsum_err=ecs)
        rows_out.append(row)
        print(row, flush=True)
check(lib.marl_tune(b"g3_tn_variant", 0))
json.dump(rows_out, open("../gpurun_out/g3_tn_lab.json", "w"), indent=1)
