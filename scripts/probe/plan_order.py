#!/usr/bin/env python3
"""Identical plans, different speeds?  N identical FusedPlans of one shape in ONE process, their temporal kernels timed
(HIP events inside afhip_plan_run) in several launch orders:

    forward    0 1 2 .. N-1 per round          reversed   N-1 .. 0 per round
    alone      each plan 12 times in a row     gap        forward, with a host sleep before every launch

If a plan INSTANCE is slow whatever the order, its placement (workspace / table addresses) matters; if the POSITION in
the rotation decides, it is what ran before (clocks, caches).  Written for the configs[3] shape, where `scripts/r03_arms.py`
showed 3.09 vs 2.82 ms between arms whose launches are identical (profiles/r03_plan_order_probe.txt).

    python scripts/probe/plan_order.py --plan c4 --dtype f32 --T 91615 --ny 180 --nx 288 --spd 1 --periods 251 --regions 3600
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from aggfly_amd import hip, synth  # noqa: E402
from r03_arms import columns  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--ny", type=int, default=180)
    ap.add_argument("--nx", type=int, default=288)
    ap.add_argument("--T", type=int, default=91615)
    ap.add_argument("--plan", default="c4")
    ap.add_argument("--spd", type=int, default=1)
    ap.add_argument("--periods", type=int, default=251)
    ap.add_argument("--regions", type=int, default=3600)
    ap.add_argument("--n", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=9)
    ap.add_argument("--short", action="store_true", help="forward and reversed series only (for counter passes)")
    ap.add_argument("--dummy-mb", type=int, default=0, help="a sacrificial HBM allocation of this size made after the cube, before any plan")
    ap.add_argument("--ws", default="torch", choices=["torch", "library", "mix"], help="where the plans' scratch comes from (see wsmode below)")
    a = ap.parse_args()
    dt = torch.float64 if a.dtype == "f64" else torch.float32
    C = a.ny * a.nx
    g = torch.Generator(device="cuda").manual_seed(1)
    cube = torch.empty((a.T, a.ny, a.nx), dtype=dt, device="cuda")
    lat = torch.linspace(0.6, 1.4, a.ny, device="cuda", dtype=torch.float32)[None, :, None]
    for k0 in range(0, a.T, 512):
        k1 = min(a.T, k0 + 512)
        noise = torch.randn((k1 - k0, a.ny, a.nx), generator=g, device="cuda", dtype=torch.float32)
        k = torch.arange(k0, k1, device="cuda", dtype=torch.float32)
        base = 15.0 + 12.0 * torch.sin(2 * np.pi * torch.floor(k / a.spd) / 365.0) + 6.0 * torch.sin(2 * np.pi * (k % a.spd) / a.spd - np.pi / 2)
        cube[k0:k1] = (base[:, None, None] * lat + 3.0 * noise).to(dt)
    ib = synth.hourly_bounds(a.T, a.spd)
    G1 = len(ib) - 1
    ob = np.round(np.linspace(0, G1, a.periods + 1)).astype(np.int64)
    cols, single = columns(a.plan)
    if single:
        ib, ob = ib[ob], np.arange(a.periods + 1, dtype=np.int64)
    wdf = synth.weights_table(a.ny, a.nx, a.regions, seed=7)
    R = int(wdf["index_right"].max()) + 1
    csr = hip.CSR(wdf["index_right"].to_numpy(), wdf["cell_id"].to_numpy(), wdf["weight"].to_numpy(), R, C)
    code = hip.F64 if a.dtype == "f64" else hip.F32
    dummy = torch.empty(a.dummy_mb << 20, dtype=torch.uint8, device="cuda") if a.dummy_mb else None
    plans = [hip.FusedPlan(a.T, C, code, ib, ob, cols) for _ in range(a.n)]
    print(plans[0].describe(), flush=True)
    # --ws torch (round 4's default of the Python host): every plan's scratch is a block of torch's caching allocator, handed to the
    # library as a caller-owned workspace; library: the library's own hipMalloc (round 3, and what a bare C caller gets); mix: even
    # plans library, odd plans torch, in one process
    def wsmode(i):
        return {"torch": None, "library": "library"}.get(a.ws, "library" if i % 2 == 0 else None)
    print("scratch: " + ", ".join(f"plan{i}={'library' if wsmode(i) else 'torch'}" for i in range(a.n)), flush=True)
    outs = [p.run(cube, csr, workspace=wsmode(i)) for i, p in enumerate(plans)]          # warm: workspaces allocated, results buffers kept
    torch.cuda.synchronize()

    def series(order, rounds, sleep=0.0):
        ms = {i: [] for i in range(a.n)}
        for _ in range(rounds):
            for i in order:
                if sleep:
                    time.sleep(sleep)
                ms[i].append(plans[i].run(cube, csr, timed=True, out=outs[i], workspace=wsmode(i))["kernel_ms"][0])
        return ms

    def show(name, ms, order):
        print(f"{name:<28}" + "  ".join(f"plan{i}: {np.median(ms[i]):.3f}" for i in order), flush=True)

    fwd = list(range(a.n))
    show("forward (launch order ->)", series(fwd, a.rounds), fwd)
    show("reversed (launch order ->)", series(fwd[::-1], a.rounds), fwd[::-1])
    if a.short:
        return
    show("forward again", series(fwd, a.rounds), fwd)
    for i in fwd:
        ms = series([i], 12)[i]
        print(f"plan{i} alone, 12 in a row:    " + " ".join(f"{v:.3f}" for v in ms), flush=True)
    show("forward, 20 ms sleep before each", series(fwd, 5, sleep=0.02), fwd)
    # untimed back-to-back launches (no host sync between them), HIP-event pairs around the temporal kernel: bench.py's method
    for i in fwd[:2]:
        plans[i].profile_begin(20)
        for _ in range(20):
            plans[i].run(cube, csr, out=outs[i], workspace=wsmode(i))
        torch.cuda.synchronize()
        ms = plans[i].profile_end()
        print(f"plan{i} 20 launches, no sync:  " + " ".join(f"{v:.3f}" for v in ms), flush=True)
    placement(plans, cube, csr, outs)
    offsets(plans, cube, csr, outs)


def placement(a_plans, cube, csr, outs):
    """the same plans with CALLER workspaces (torch tensors): is it the plan-owned workspace's placement?"""
    for i in (0, 1):
        p = a_plans[i]
        for trial in range(3):
            ws = torch.empty(p.workspace_bytes(csr), dtype=torch.uint8, device="cuda")
            ms = [p.run(cube, csr, timed=True, out=outs[i], workspace=ws)["kernel_ms"][0] for _ in range(8)]
            print(f"plan{i} caller workspace #{trial} @0x{ws.data_ptr():x}: median {np.median(ms):.3f}", flush=True)
            del ws
        ms = [p.run(cube, csr, timed=True, out=outs[i], workspace="library")["kernel_ms"][0] for _ in range(8)]
        print(f"plan{i} library-owned scratch again:    median {np.median(ms):.3f}", flush=True)


def offsets(a_plans, cube, csr, outs):
    """a caller workspace at different offsets inside ONE larger buffer: does the address itself matter?"""
    p = a_plans[1]
    need = p.workspace_bytes(csr)
    big = torch.empty(need + (1 << 30), dtype=torch.uint8, device="cuda")
    print(f"cube @0x{cube.data_ptr():x} ({cube.numel() * cube.element_size()} bytes), big buffer @0x{big.data_ptr():x}", flush=True)
    for off in (0, 256, 4096, 65536, 1 << 20, 2 << 20, 3 << 20, 16 << 20, 64 << 20, 256 << 20, 512 << 20, (1 << 30) - 4096):
        ws = big[off:off + need]
        ms = [p.run(cube, csr, timed=True, out=outs[1], workspace=ws)["kernel_ms"][0] for _ in range(6)]
        print(f"  workspace at +{off:>11}: median {np.median(ms):.3f}", flush=True)


if __name__ == "__main__":
    main()
