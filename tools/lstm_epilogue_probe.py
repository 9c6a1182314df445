import os, sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import torch as th
from g3_lab import image, padded, timeit, p4, lib, check, dev
m, n, nin = 4096, 256, 368
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
"""How much of the fused LSTM launch is its epilogue: the same launch without the activated-gate stores (33.5 MB per
step at C3) and / or without the image of h' (6.3 MB + the LDS transposition and the three-way split)."""
cases = (("all outputs", gt.data_ptr(), h3n.data_ptr()), ("no gates", None, h3n.data_ptr()), ("gates, no image", gt.data_ptr(), None),
         ("no gates, no image", None, None))
for rep in range(3):
  for name, gp, ip in cases:
    call = lambda: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(), bias.data_ptr(), cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(), gp, ip, m, n, p4(n), p4(4 * n), 2, 2, None))
    print(rep, name, round(timeit(call, 100), 1), "us (two cells)", flush=True)
