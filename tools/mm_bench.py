import torch, time
dev = "cuda"
def bench(m, n, k, tn=False):
    if tn:
        a = torch.randn(k, m, device=dev); b = torch.randn(k, n, device=dev)
        f = lambda: a.t() @ b
    else:
        a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev)
        f = lambda: a @ b.t()
    for _ in range(5): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{'TN' if tn else 'NT'} m={m} n={n} k={k}: {dt*1e6:.1f} us  {2*m*n*k/dt/1e12:.1f} TF")
torch.backends.cuda.matmul.allow_tf32 = False
bench(4096, 2048, 640)
bench(65536, 640, 2048)
bench(65536, 384, 256)
bench(4096, 256, 1024)
bench(2048, 640, 65536, tn=True)
bench(384, 256, 65536, tn=True)
bench(4096, 4096, 4096)
