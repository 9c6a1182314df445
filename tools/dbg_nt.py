import os, sys, torch as th
sys.path.insert(0, os.getcwd())
from marlclassification_amd import _lib
lib, check = _lib.load(), _lib.check
dev = th.device("cuda:0")
for (m, n, k) in [(128, 128, 32), (128, 128, 64), (128, 128, 128), (128, 64, 160), (256, 512, 96), (4096, 256, 1024), (95, 45, 24)]:
    g = th.Generator().manual_seed(1)
    a = th.randn(m, k, generator=g); b = th.randn(n, k, generator=g)
    k4 = (k + 3) & ~3
    ws = th.zeros(m * k4 + 64, device=dev); ad = ws[: m * k4].view(m, k4); ad[:, :k] = a.to(dev)
    bd = th.zeros(n, k4, device=dev); bd[:, :k] = b.to(dev)
    c = th.zeros(m, n, device=dev)
    img = th.zeros(lib.marl_gemm_weight_image_bytes(n, k) // 4 + 64, device=dev)
    check(lib.marl_gemm_nt_weights(ad.data_ptr(), k4, bd.data_ptr(), k4, None, c.data_ptr(), n, m, n, k, 0, img.data_ptr(), None))
    th.cuda.synchronize()
    ref = a @ b.t()
    err = (c.cpu() - ref).abs()
    print(m, n, k, "max err", err.max().item(), flush=True)
    if err.max() > 1e-3:
        for t in range((k + 31) // 32):
            part = a[:, t*32:(t+1)*32] @ b[:, t*32:(t+1)*32].t()
            print("   tile", t, "corr:", ((ref - c.cpu()) * part).sum().item() / (part * part).sum().item())
