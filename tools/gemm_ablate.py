import ctypes as C, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch as th
from marlclassification_amd import _lib
lib=_lib.load()
dev=th.device("cuda:0")
m,n,k=65536,384,2048
a=th.randn(m,k,device=dev); b=th.randn(n,k,device=dev); c=th.zeros(m,n,device=dev)
for flags in (0,1,2,3,4,7,8,15):
    lib.marl_debug_set_gemm_flags(flags)
    for _ in range(2): lib.marl_gemm_nt(a.data_ptr(),k,b.data_ptr(),k,None,c.data_ptr(),n,m,n,k,0,None)
    th.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): lib.marl_gemm_nt(a.data_ptr(),k,b.data_ptr(),k,None,c.data_ptr(),n,m,n,k,0,None)
    th.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print(f"flags={flags:2d}: {dt*1e6:8.1f} us  {2.0*m*n*k/dt/1e12:6.1f} TF-equiv")
