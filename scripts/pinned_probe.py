import torch, time
for mb in (64, 128, 256, 512, 1024, 2048):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8, pin_memory=True); h.fill_(1)
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    # copy in 128 MB pieces
    t0 = time.perf_counter()
    for o in range(0, n, 128 << 20):
        d[o:o + (128 << 20)].copy_(h[o:o + (128 << 20)], non_blocking=True)
    torch.cuda.synchronize(); dt2 = time.perf_counter() - t0
    print(mb, "MiB pinned: one copy %.1f GB/s, 128-MiB pieces %.1f GB/s" % (n / 1e9 / best, n / 1e9 / dt2))
