#!/usr/bin/env python3
"""Page-locked host -> HBM copy rate of the box (one stream, and two streams at once), for sizing the ingest pipeline."""
import time, torch
for mb in (16, 64, 256, 1024):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device="cuda")
    d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.cuda.stream(s1): d[:n // 2].copy_(h[:n // 2], non_blocking=True)
    with torch.cuda.stream(s2): d[n // 2:].copy_(h[n // 2:], non_blocking=True)
    torch.cuda.synchronize(); two = time.perf_counter() - t0
    print(f"{mb:5d} MB: {n / best / 1e9:6.1f} GB/s one stream, {n / two / 1e9:6.1f} GB/s split over two streams", flush=True)
