import os, sys, torch as th
sys.path.insert(0, os.getcwd())
from marlclassification_amd import _lib
lib, check = _lib.load(), _lib.check
dev = th.device("cuda:0")
m, n, k = 128, 64, 64
for case in ("ones", "a_tile1_only", "b_tile1_only"):
    a = th.ones(m, k); b = th.ones(n, k)
    if case == "a_tile1_only": a[:, :32] = 0; a[:, 32:] = th.arange(32).float() + 1
    if case == "b_tile1_only": b[:, :32] = 0; b[:, 32:] = th.arange(32).float() + 1
    ad, bd = a.to(dev), b.to(dev); c = th.zeros(m, n, device=dev)
    check(lib.marl_gemm_nt(ad.data_ptr(), k, bd.data_ptr(), k, None, c.data_ptr(), n, m, n, k, 0, None))
    th.cuda.synchronize()
    ref = a @ b.t()
    print(case, "ref", ref[0, 0].item(), "got", c[0, :4].cpu().tolist(), c[64:66, 32:34].cpu().tolist())
