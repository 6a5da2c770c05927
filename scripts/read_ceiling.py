#!/usr/bin/env python3
"""Read-only bandwidth reference on this box: torch's own reductions over the same 21.7 GB
cube (what a tuned library kernel reaches) beside the fused temporal kernel."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from aggfly_amd import hip, synth

T, ny, nx = 8760, 215, 1440
cube = bench.make_cube(torch, T, ny, nx, torch.float64, 1)
gb = cube.numel() * 8 / 1e9
out = {}
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); e.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))
flat = cube.view(-1)
out["torch_sum_all_GBps"] = gb / timeit(lambda: flat.sum()) * 1e3
out["torch_sum_time_axis_GBps"] = gb / timeit(lambda: cube.sum(dim=0)) * 1e3
out["torch_max_all_GBps"] = gb / timeit(lambda: flat.max()) * 1e3
dst = torch.empty_like(cube[: T // 2])
half = cube[: T // 2]
out["torch_copy_half_GBps(read+write)"] = 2 * half.numel() * 8 / 1e9 / timeit(lambda: dst.copy_(half)) * 1e3
ib = synth.hourly_bounds(T); ob = np.array([0, len(ib) - 1])
tab = synth.weights_table(ny, nx, 3100, seed=7); R = int(tab.index_right.max()) + 1
csr = hip.CSR(tab.index_right.to_numpy(), tab.cell_id.to_numpy(), tab.weight.to_numpy(), R, ny * nx)
for name, cols in (("fused_mean_only", [dict(inner="mean", outer="sum")]), ("fused_c2", bench.c2_columns())):
    plan = hip.FusedPlan(T, ny * nx, hip.F64, ib, ob, cols)
    ms = [plan.run(cube, csr, timed=True)["kernel_ms"][0] for _ in range(6)][1:]
    out[name + "_GBps"] = gb / float(np.median(ms)) * 1e3
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/read_ceiling.json", "w"), indent=1)
