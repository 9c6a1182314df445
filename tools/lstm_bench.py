"""Times the NT products of the C3 iteration that have registered weights as B (bf16x6 path):
runs bench-like forward steps through the engine and reports class timings via the profile hook."""
import ctypes as C, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import bench
from marlclassification_amd import _lib
from marlclassification_amd.fused import FusedA2C, draw_episode_device
from marlclassification_amd.networks import ModelsWrapper
from marlclassification_amd.networks.vision import CNN_BY_NAME
lib = _lib.load()
dev = th.device("cuda:0")
C3 = bench.C3
th.manual_seed(0)
model = ModelsWrapper(CNN_BY_NAME[C3["ft_extr"]](C3["window"]), C3["n_b"], C3["n_a"], C3["n_m"], C3["n_m_o"], C3["n_d"], 2, 4,
                      C3["nb_class"], C3["nlb"], C3["nla"]).to(dev)
flat = model.flat_state()
eng = model.hip_engine([[1, 0], [-1, 0], [0, 1], [0, -1]])
eng.configure(16, 256, 16, (3, 256, 256))
fa = FusedA2C(eng, flat, 1e-4, 0.99)
img = th.rand(256, 3, 256, 256, device=dev)
y = th.randint(0, 45, (256,), device=dev)
for it in range(3):
    fa.iteration(img, y, draw_episode_device(eng, 1, it))
th.cuda.synchronize()
res = {}
for c in range(3):
    lib.marl_profile_begin(c, 4096)
    fa.iteration(img, y, draw_episode_device(eng, 1, 10 + c))
    tot, cnt = C.c_double(0), C.c_int(0)
    lib.marl_profile_end(C.byref(tot), C.byref(cnt))
    res[c] = (round(tot.value, 3), cnt.value)
print("class ms (launches): lstm", res[0], "nt", res[1], "tn", res[2])
