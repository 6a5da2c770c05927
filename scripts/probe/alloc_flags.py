#!/usr/bin/env python3
"""Does an allocation flag steer the bin-count kernel's slow / fast mode (configs[3] shape: a 19 GB cube; profiles/r04_plan_order_probe.txt: the mode is drawn per
allocation of the CUBE)?  The cube is allocated by hipMalloc (through torch), hipExtMallocWithFlags(default / fine-grained / uncached / contiguous) and
hipMallocManaged-free variants, several times each, filled the same way, and the same plan is timed on each.

    python scripts/probe/alloc_flags.py [--trials 5]"""
import argparse, ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aggfly_amd import hip, synth

libhip = C.CDLL("libamdhip64.so")


class Raw:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}


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
    ap.add_argument("--trials", type=int, default=5)
    a = ap.parse_args()
    T, ny, nx = 91615, 180, 288
    Cc = ny * nx
    edges = np.arange(-20, 50, 5.0)
    cols = [dict(inner="bins", inner_args=(edges[i], edges[i + 1], 0)) for i in range(13)]
    ib = np.round(np.linspace(0, T, 252)).astype(np.int64); ob = np.arange(252, dtype=np.int64)
    tab = synth.weights_table(ny, nx, 3600, seed=7, secondary=True)
    R = int(tab["index_right"].max()) + 1
    csr = hip.CSR(tab["index_right"].to_numpy(), tab["cell_id"].to_numpy(), tab["weight"].to_numpy(), R, Cc)
    plan = hip.FusedPlan(T, Cc, hip.F32, ib, ob, cols)

    def time_on(cube, n=10):
        out = plan.run(cube, csr)
        for _ in range(3):
            plan.run(cube, csr, out=out)
        return float(np.median([plan.run(cube, csr, timed=True, out=out)["kernel_ms"][0] for _ in range(n)])), float(np.median(hip.read_probe(cube, 6)))

    nbytes = T * Cc * 4
    for name, flag in (("torch (hipMalloc)", None), ("hipExtMallocWithFlags default", 0), ("... contiguous", 4), ("... uncached", 3), ("... fine-grained", 1), ("torch (hipMalloc)", None)):
        times = []
        for t in range(a.trials):
            if flag is None:
                cube = torch.empty((T, ny, nx), dtype=torch.float32, device="cuda")
                ptr = None
            else:
                p = C.c_void_p()
                rc = libhip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(nbytes), C.c_uint(flag))
                if rc != 0:
                    print(f"{name}: hipExtMallocWithFlags -> error {rc}", flush=True)
                    break
                ptr = p.value
                cube = torch.as_tensor(Raw(ptr, T * Cc), device="cuda").view(T, ny, nx)
                assert cube.data_ptr() == ptr
            fill(cube)
            k_ms, r_ms = time_on(cube)
            times.append((k_ms, r_ms))
            del cube
            torch.cuda.synchronize()
            if ptr is not None:
                libhip.hipFree(C.c_void_p(ptr))
            else:
                torch.cuda.empty_cache()
        print(f"{name:32s} kernel ms " + " ".join(f"{k:.3f}" for k, _ in times) + "   | bare read ms " + " ".join(f"{r:.3f}" for _, r in times), flush=True)


if __name__ == "__main__":
    main()
