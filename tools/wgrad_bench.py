"""Micro-benchmark + check of the conv weight-gradient kernel (marl_cnn_wgrad) on the RESISC45
layer shapes, with in-process A/B over the tuning knobs (marl_tune).
usage: python tools/wgrad_bench.py [rows] [knob=value,...;knob=value,... ...]"""
import ctypes as C
import os
import sys

import torch as th
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marlclassification_amd import _lib  # noqa: E402

lib = _lib.load()
dev = th.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variants = [v for v in (sys.argv[2].split(";") if len(sys.argv) > 2 else [""])]
LAYERS = [  # cin, cout, hin, groups of the layer below, first
    (3, 16, 12, 1, True), (16, 32, 6, 2, False), (32, 64, 3, 4, False)]
if os.environ.get("WGRAD_SET") == "aid32":  # BASELINE configs[4]: AidCnn on 32 x 32 patches (deeper layers only)
    LAYERS = [(16, 32, 16, 2, False), (32, 64, 8, 4, False), (64, 128, 4, 8, False)]
if os.environ.get("WGRAD_SET") == "aid24":  # configs[3]: f = 24
    LAYERS = [(16, 32, 12, 2, False), (32, 64, 6, 4, False), (64, 128, 3, 8, False)]
nb, H, W = 256, 256, 256
g = th.Generator(device=dev).manual_seed(0)


def run(layer, check):
    cin, cout, hin, G, first = layer
    hout = (hin - 1) // 2 + 1
    P = hout * hout
    dz = th.randn(rows, P, cout, device=dev, generator=g)
    img = th.rand(nb, 3, H, W, device=dev, generator=g)
    pos = th.stack([th.randint(H - hin, (rows,), device=dev, generator=g),
                    th.randint(W - hin, (rows,), device=dev, generator=g)], -1).int().contiguous()
    zin = th.randn(rows, hin * hin, cin, device=dev, generator=g)
    cpg = max(1, cin // G)
    zz = zin.view(rows, hin * hin, G, cpg)
    mean = zz.mean(dim=(1, 3))
    rstd = 1.0 / th.sqrt(zz.var(dim=(1, 3), unbiased=False) + 1e-5)
    gst = th.stack([mean, rstd], -1).contiguous()
    gamma = 1 + 0.1 * th.randn(cin, device=dev, generator=g)
    beta = 0.1 * th.randn(cin, device=dev, generator=g)
    dw = th.empty(cout, 9 * cin, device=dev)
    db = th.empty(cout, device=dev)
    sb = lib.marl_cnn_wgrad_scratch(rows, cin, cout, hin, G, int(first))
    scratch = th.empty(sb // 4 + 64, device=dev)
    st = th.cuda.current_stream().cuda_stream

    def call():
        rc = lib.marl_cnn_wgrad(dz.data_ptr(), img.data_ptr(), 0, pos.data_ptr(),
                                None if first else zin.data_ptr(), gst.data_ptr(), gamma.data_ptr(),
                                beta.data_ptr(), rows, nb, 3, H, W, cin, cout, hin, G, dw.data_ptr(),
                                db.data_ptr(), scratch.data_ptr(), scratch.numel() * 4, st)
        assert rc == 0, lib.marl_last_error()

    call()
    th.cuda.synchronize()
    if check:
        n = min(rows, 2048)
        if first:
            ar = th.arange(hin, device=dev)
            x = th.stack([img[r % nb, :cin][:, pos[r, 0] + ar][:, :, pos[r, 1] + ar] for r in range(n)])
        else:
            xh = (zz[:n] - mean[:n, None, :, None]) * rstd[:n, None, :, None]
            x = F.silu(xh.reshape(n, hin * hin, cin) * gamma + beta).view(n, hin, hin, cin).permute(0, 3, 1, 2)
        wt = th.zeros(cout, cin, 3, 3, device=dev, requires_grad=True)
        y = F.conv2d(x.double(), wt.double(), stride=2, padding=1)
        gz = dz[:n].view(n, hout, hout, cout).permute(0, 3, 1, 2).double()
        (y * gz).sum().backward()
        # reference [co][ci][kh][kw] -> [co][tap*cin+ci]
        ref = wt.grad.permute(0, 2, 3, 1).reshape(cout, 9 * cin).float()
        sub = rows
        if n < rows:  # re-run on the checked prefix only
            rc = lib.marl_cnn_wgrad(dz.data_ptr(), img.data_ptr(), 0, pos.data_ptr(),
                                    None if first else zin.data_ptr(), gst.data_ptr(), gamma.data_ptr(),
                                    beta.data_ptr(), n, nb, 3, H, W, cin, cout, hin, G, dw.data_ptr(),
                                    db.data_ptr(), scratch.data_ptr(), scratch.numel() * 4, st)
            assert rc == 0
        th.cuda.synchronize()
        err = (dw - ref).abs().max().item() / ref.abs().max().item()
        berr = (db - dz[:n].sum(dim=(0, 1))).abs().max().item() / db.abs().max().item()
        print(f"  check layer cin={cin} cout={cout}: dw rel err {err:.2e}, db rel err {berr:.2e}")
        assert err < 1e-4 and berr < 1e-4
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    for _ in range(3):
        call()
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    th.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    flops = 2.0 * rows * P * cout * 9 * cin
    return us, flops / us / 1e6


for rnd in range(2):
    for v in variants:
        for kv in filter(None, v.split(",")):
            k, val = kv.split("=")
            lib.marl_tune(k.encode(), int(val))
        res = [run(l, rnd == 0 and v == variants[0]) for l in LAYERS]
        print(f"round {rnd} [{v or 'default'}]: " + "  ".join(f"L{i}: {us:7.1f} us {tf:5.1f} TF" for i, (us, tf) in enumerate(res)))
