"""Throughput of the product's Trainer.train_epoch (the loop `train` runs) next to bench.py's iteration:
C3 shapes, uint8 images resident in HBM, 40 iterations.  usage: python tools/train_loop_bench.py"""
import sys
import time
import os

import torch as th

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import C3, IMG, NA, NS  # noqa: E402
from marlclassification_amd.networks.vision import CNN_BY_NAME  # noqa: E402
from marlclassification_amd.core import Environment, EpisodeSampler, MultiAgent  # noqa: E402
from marlclassification_amd.networks import ModelsWrapper  # noqa: E402
from marlclassification_amd.training import Trainer  # noqa: E402

dev = th.device("cuda", 0)
actions = [[1, 0], [-1, 0], [0, 1], [0, -1]]
th.manual_seed(0)
model = ModelsWrapper(CNN_BY_NAME[C3["ft_extr"]](C3["window"]), C3["n_b"], C3["n_a"], C3["n_m"], C3["n_m_o"],
                      C3["n_d"], 2, len(actions), C3["nb_class"], C3["nlb"], C3["nla"]).to(dev)
sampler = EpisodeSampler(MultiAgent(NA, model), Environment(actions, C3["window"]), NS)
trainer = Trainer(model, C3["nb_class"], 1e-4, 0.99)
nb, iters = 256, 40
batches = [(th.randint(0, 256, (nb, *IMG), dtype=th.uint8, device=dev), th.randint(0, C3["nb_class"], (nb,), device=dev))
           for _ in range(4)]
loader = [batches[i % 4] for i in range(iters)]
trainer.train_epoch(loader[:5], 0, sampler)
th.cuda.synchronize()
t0 = time.perf_counter()
trainer.train_epoch(loader, 1, sampler)
t_host = time.perf_counter() - t0  # the host is done enqueueing here
th.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"host enqueue time {1e3 * t_host / iters:.3f} ms / iteration")
print(f"Trainer.train_epoch: {1e3 * dt / iters:.3f} ms / iteration, {nb * NA * NS * iters / dt / 1e6:.2f} M agent-env-steps/s")
