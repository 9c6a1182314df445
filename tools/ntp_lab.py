"""Round 6: the phase-pipelined NT / LSTM kernels (gemm_nt3p_kernel) against the round-5 plans on the C3 shapes.
LSTM launch (two cells, R = 4096): variants 2 (128-row, shipped) / 5 (256 x 128, 8 waves) / 6 (256 x 128, 4 waves);
NT: dU [65536 x 256 x 2048] and the heads [65536 x 384 x 256] x 2, variants 2 / 21 / 22; the LSTM-like [4096 x 1024 x 624].
python tools/ntp_lab.py"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th  # noqa: E402
from g3_lab import image, padded, timeit, p4, lib, check, dev  # noqa: E402

out = []
m, n, nin = 4096, 256, 368
g = th.Generator().manual_seed(1)
u, h, cprev = th.randn(m, nin, generator=g), th.randn(m, n, generator=g), th.randn(m, n, generator=g)
wih, whh = th.randn(4 * n, nin, generator=g) / nin ** 0.5, th.randn(4 * n, n, generator=g) / n ** 0.5
bias = th.randn(4 * n, generator=g).to(dev)
u3, h3 = image(padded(u.to(dev), p4(nin)), nin), image(padded(h.to(dev), p4(n)), n)
wih3, whh3 = image(padded(wih.to(dev), p4(nin)), nin), image(padded(whh.to(dev), p4(n)), n)
cpd = padded(cprev.to(dev), p4(n))
res = {}
for rep in range(3):
    for variant in (2, 5, 6):
        hn, cn = th.zeros(m, p4(n), device=dev), th.zeros(m, p4(n), device=dev)
        gt = th.zeros(m, p4(4 * n), device=dev)
        h3n = th.zeros(lib.marl_image_bytes(m, n) + 256, dtype=th.uint8, device=dev)
        call = lambda cells: check(lib.marl_lstm_images(u3.data_ptr(), nin, h3.data_ptr(), wih3.data_ptr(), whh3.data_ptr(), bias.data_ptr(),
                                                        cpd.data_ptr(), hn.data_ptr(), cn.data_ptr(), gt.data_ptr(), h3n.data_ptr(), m, n,
                                                        p4(n), p4(4 * n), variant, cells, None))
        call(1)
        th.cuda.synchronize()
        r = (hn.clone(), cn.clone(), gt.clone(), h3n.clone())
        res.setdefault(variant, r)
        same = all(th.equal(a, b) for a, b in zip(r, res[2]))
        out.append(dict(kind="lstm", variant=variant, us_two_cells=round(timeit(lambda: call(2), 100), 1), bit_equal_to_variant_2=same))
        print(out[-1], flush=True)

for (mm, nn, kk, segs) in ((65536, 256, 1024, 2), (65536, 384, 256, 1), (4096, 1024, 624, 1)):
    g = th.Generator().manual_seed(mm + nn + kk)
    ad = padded(th.randn(mm, kk, generator=g).to(dev), p4(kk))
    bd = padded((th.randn(nn, kk, generator=g) / kk ** 0.5).to(dev), p4(kk))
    a3, b3 = image(ad, kk), image(bd, kk)
    ref = None
    for rep in range(2):
        for variant in (2, 21, 22):
            c1 = th.zeros(mm, p4(nn), device=dev)
            if segs == 1:
                fn = lambda: check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), None, c1.data_ptr(), c1.shape[1], mm, nn, kk, 0, variant, None))
            else:  # two products accumulate = the two-segment dU product (same operand twice)
                def fn():
                    check(lib.marl_gemm_nt_images(a3.data_ptr(), b3.data_ptr(), None, c1.data_ptr(), c1.shape[1], mm, nn, kk, 0, variant, None))
            fn()
            th.cuda.synchronize()
            if ref is None:
                ref = c1.clone()
            us = timeit(fn, 20)
            out.append(dict(kind="nt", m=mm, n=nn, k=kk, variant=variant, us=round(us, 1), tf=round(2.0 * mm * nn * kk / us / 1e6, 1),
                            bit_equal_to_variant_2=bool(th.equal(c1, ref))))
            print(out[-1], flush=True)
json.dump(out, open("gpurun_out/ntp_lab.json", "w"), indent=1)
