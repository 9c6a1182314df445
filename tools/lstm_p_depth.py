"""Round 6: is the phase-pipelined LSTM loop bound by latency x bytes in flight or by bandwidth?  The 4-wave 256 x 128
form with a 4-stage (variant 6) and a 3-stage ring (variant 7, ablation build only): per-step time from two K lengths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from marlclassification_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libmarl_abl.so")
import torch as th
from g3_lab import image, padded, timeit, p4, lib, check, dev
m, n = 4096, 256
res = {}
for nin in (368, 16):
    g = th.Generator().manual_seed(1)
    u, h, cprev = th.randn(m, nin, generator=g), th.randn(m, n, generator=g), th.randn(m, n, generator=g)
    wih, whh = th.randn(4 * n, nin, generator=g) / nin ** 0.5, th.randn(4 * n, n, generator=g) / n ** 0.5
    bias = th.randn(4 * n, generator=g).to(dev)
    u3, h3 = image(padded(u.to(dev), p4(nin)), nin), image(padded(h.to(dev), p4(n)), n)
    wih3, whh3 = image(padded(wih.to(dev), p4(nin)), nin), image(padded(whh.to(dev), p4(n)), n)
    cpd = padded(cprev.to(dev), p4(n))
    hn, cn = th.zeros(m, p4(n), device=dev), th.zeros(m, p4(n), device=dev)
    for rep in range(3):
        for variant in (6, 7, 2):
            call = lambda: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(), bias.data_ptr(), cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(), None, None, m, n, p4(n), p4(4 * n), variant, 2, None))
            us = timeit(call, 100)
            res.setdefault((variant, nin), []).append(us)
for v in (6, 7, 2):
    a, b = min(res[(v, 368)]), min(res[(v, 16)])
    print(f"variant {v}: 39 steps {a:.1f} us, 17 steps {b:.1f} us -> {(a - b) / 22:.3f} us per step, intercept {b - 17 * (a - b) / 22:.1f} us")
