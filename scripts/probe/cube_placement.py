#!/usr/bin/env python3
"""Is the 7-9 % slow mode of the bin-count kernel (configs[3] shape) a property of WHERE THE CUBE LANDED?

profiles/r03_plan_order_probe.txt saw three states per process (all plans fast / only the first plan slow / all plans slow) and
round 4's first probe (profiles/r04_plan_order_probe.txt) saw whole processes fast or slow whatever the scratch's allocator.
Here ONE process allocates the cube several times — freeing it, shifting the next allocation with dummies of different sizes,
carving it out of one large block at different offsets — and times the same plan on every placement.

    python scripts/probe/cube_placement.py [--trials 8]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

import torch  # noqa: E402

from aggfly_amd import hip, synth  # noqa: E402


def fill(cube, seed=1):
    g = torch.Generator(device="cuda").manual_seed(seed)
    T = cube.shape[0]
    for k0 in range(0, T, 4096):
        k1 = min(T, k0 + 4096)
        k = torch.arange(k0, k1, device="cuda", dtype=torch.float32)
        base = 15.0 + 12.0 * torch.sin(2 * np.pi * k / 365.0)
        cube[k0:k1] = base[:, None, None] + 3.0 * torch.randn((k1 - k0,) + tuple(cube.shape[1:]), generator=g, device="cuda", dtype=torch.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=8)
    ap.add_argument("--T", type=int, default=91615)
    ap.add_argument("--ny", type=int, default=180)
    ap.add_argument("--nx", type=int, default=288)
    a = ap.parse_args()
    T, ny, nx = a.T, a.ny, a.nx
    C = ny * nx
    edges = np.arange(-20, 50, 5.0)
    cols = [dict(inner="bins", inner_args=(edges[i], edges[i + 1], 0)) for i in range(13)]
    ib = np.round(np.linspace(0, T, 252)).astype(np.int64)
    ob = np.arange(252, dtype=np.int64)
    tab = synth.weights_table(ny, nx, 3600, seed=7, secondary=True)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, C)
    plan = hip.FusedPlan(T, C, hip.F32, ib, ob, cols)
    print(plan.describe(), flush=True)

    def time_on(cube, n=10):
        out = plan.run(cube, csr)
        for _ in range(3):
            plan.run(cube, csr, out=out)
        ms = [plan.run(cube, csr, timed=True, out=out)["kernel_ms"][0] for _ in range(n)]
        return float(np.median(ms))

    rng = np.random.default_rng(0)
    print("## fresh allocations of the cube (freed in between; odd trials keep a dummy of random size alive to shift the address)", flush=True)
    keep = []
    for t in range(a.trials):
        cube = torch.empty((T, ny, nx), dtype=torch.float32, device="cuda")
        fill(cube)
        print(f"trial {t}: cube @0x{cube.data_ptr():x}  (mod 2 MiB: {cube.data_ptr() % (2 << 20):>8}, mod 1 GiB: {cube.data_ptr() % (1 << 30) >> 20:>5} MiB)  "
              f"temporal kernel {time_on(cube):.3f} ms", flush=True)
        del cube
        torch.cuda.empty_cache()
        if t % 2 == 0:
            keep.append(torch.empty(int(rng.integers(64, 4096)) << 20, dtype=torch.uint8, device="cuda"))
    del keep
    torch.cuda.empty_cache()
    print("## one 24 GiB block, the cube carved out at different offsets", flush=True)
    nbytes = T * C * 4
    big = torch.empty(nbytes + (4 << 30), dtype=torch.uint8, device="cuda")
    for off in (0, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 64 << 20, 1 << 30, (1 << 30) + (1 << 20), 3 << 30):
        cube = big[off:off + nbytes].view(torch.float32).view(T, ny, nx)
        fill(cube)
        print(f"offset {off:>11}: cube @0x{cube.data_ptr():x}  temporal kernel {time_on(cube):.3f} ms", flush=True)
    del big, cube
    torch.cuda.empty_cache()
    print("## hipMalloc'ed by the library's scratch path vs torch: the plan's scratch from each allocator on one cube", flush=True)
    cube = torch.empty((T, ny, nx), dtype=torch.float32, device="cuda")
    fill(cube)
    for ws in (None, "library", None, "library"):
        out = plan.run(cube, csr, workspace=ws)
        ms = [plan.run(cube, csr, timed=True, out=out, workspace=ws)["kernel_ms"][0] for _ in range(10)]
        print(f"scratch {'torch' if ws is None else 'library'}: {float(np.median(ms)):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
