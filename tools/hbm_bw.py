import torch, time
dev = torch.device("cuda:0")
for n in (2**21, 2**24, 2**26, 2**28):   # floats: 8 MB, 64 MB, 256 MB, 1 GB
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): b.copy_(a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    for _ in range(3): a.sum()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): a.sum()
    torch.cuda.synchronize(); ds = (time.perf_counter() - t0) / 20
    print(f"{n*4/2**20:7.0f} MiB: copy {2*n*4/dt/1e12:5.2f} TB/s (r+w)   sum {n*4/ds/1e12:5.2f} TB/s (read)")
