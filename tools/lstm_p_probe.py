"""Round 6: where the time of the phase-pipelined LSTM launch goes: variants 2 / 5 / 6 with and without the gate and
image stores, and with K cut to a quarter (nin = 16, n = 256 stays: 17 steps instead of 39).  python tools/lstm_p_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th
from g3_lab import image, padded, timeit, p4, lib, check, dev
m, n = 4096, 256
for nin in (368, 16):
    g = th.Generator().manual_seed(1)
    u, h, cprev = th.randn(m, nin, generator=g), th.randn(m, n, generator=g), th.randn(m, n, generator=g)
    wih, whh = th.randn(4 * n, nin, generator=g) / nin ** 0.5, th.randn(4 * n, n, generator=g) / n ** 0.5
    bias = th.randn(4 * n, generator=g).to(dev)
    u3, h3 = image(padded(u.to(dev), p4(nin)), nin), image(padded(h.to(dev), p4(n)), n)
    wih3, whh3 = image(padded(wih.to(dev), p4(nin)), nin), image(padded(whh.to(dev), p4(n)), n)
    cpd = padded(cprev.to(dev), p4(n))
    hn, cn = th.zeros(m, p4(n), device=dev), th.zeros(m, p4(n), device=dev)
    gt = th.zeros(m, p4(4 * n), device=dev)
    h3n = th.zeros(lib.marl_image_bytes(m, n) + 256, dtype=th.uint8, device=dev)
    cases = (("all outputs", gt.data_ptr(), h3n.data_ptr()), ("no gates, no image", None, None))
    for rep in range(2):
        for name, gp, ip in cases:
            for variant in (2, 5, 6):
                call = lambda: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(), bias.data_ptr(), cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(), gp, ip, m, n, p4(n), p4(4 * n), variant, 2, None))
                print(f"nin={nin} steps={(nin + 15) // 16 + 16} {name:20s} variant {variant}: {timeit(call, 100):6.1f} us (two cells)", flush=True)
