"""Experiment: does the chip overlap the latency-bound per-step kernels of one half of the batch
with the GEMMs of the other half?  Two independent B/2 trainers on two streams (no events inside
the loop: batch elements never interact) against one trainer at B.  Prints ms per B images.

usage: python tools/split_proto.py [B] [ways]
"""
import sys
import time

import torch as th

sys.path.insert(0, ".")
import bench  # noqa: E402
from marlclassification_amd.fused import FusedA2C, draw_episode_device  # noqa: E402
from marlclassification_amd.networks import ModelsWrapper  # noqa: E402
from marlclassification_amd.networks.vision import CNN_BY_NAME  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
WAYS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = th.device("cuda", 0)
C3, NA, NS, IMG = bench.C3, bench.NA, bench.NS, bench.IMG
actions = [[1, 0], [-1, 0], [0, 1], [0, -1]]


def make(nb):
    th.manual_seed(0)
    model = ModelsWrapper(CNN_BY_NAME[C3["ft_extr"]](C3["window"]), C3["n_b"], C3["n_a"], C3["n_m"],
                          C3["n_m_o"], C3["n_d"], 2, len(actions), C3["nb_class"], C3["nlb"],
                          C3["nla"]).to(dev)
    flat = model.flat_state()
    eng = model.hip_engine(actions)
    eng.configure(NA, nb, NS, IMG)
    fa = FusedA2C(eng, flat, bench.LR, bench.GAMMA)
    img = th.rand(nb, *IMG, device=dev)
    y = th.randint(0, C3["nb_class"], (nb,), device=dev)
    return model, eng, fa, img, y


def timeit(fn, n=12, warm=4):
    for _ in range(warm):
        fn()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    th.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


whole = make(B)
it = [0]


def one():
    _, eng, fa, img, y = whole
    fa.iteration(img, y, draw_episode_device(eng, 42, it[0]))
    it[0] += 1


print(f"one trainer, B={B}: {timeit(one):.3f} ms")

parts = [make(B // WAYS) for _ in range(WAYS)]
streams = [th.cuda.Stream(device=dev) for _ in range(WAYS)]


def split():
    cur = th.cuda.current_stream(dev)
    for s, (_, eng, fa, img, y) in zip(streams, parts):
        s.wait_stream(cur)
        with th.cuda.stream(s):
            fa.iteration(img, y, draw_episode_device(eng, 42, it[0]))
    for s in streams:
        cur.wait_stream(s)
    it[0] += 1


print(f"{WAYS} trainers at B={B // WAYS} on {WAYS} streams: {timeit(split):.3f} ms")


def serial():
    for (_, eng, fa, img, y) in parts:
        fa.iteration(img, y, draw_episode_device(eng, 42, it[0]))
    it[0] += 1


print(f"{WAYS} trainers at B={B // WAYS}, one stream: {timeit(serial):.3f} ms")
