import os, sys, torch as th
sys.path.insert(0, os.getcwd())
from marlclassification_amd import _lib
lib, check = _lib.load(), _lib.check
dev = th.device("cuda:0")
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
a = th.randn(m, k, device=dev); b = th.randn(n, k, device=dev); c = th.zeros(m, n, device=dev)
for i in range(8):
    check(lib.marl_gemm_nt(a.data_ptr(), k, b.data_ptr(), k, None, c.data_ptr(), n, m, n, k, 0, None))
th.cuda.synchronize()
