#!/usr/bin/env python3
"""Turns the output of scripts/r03_arms.py runs (as collected by the r04_*.sh scripts: '== args' lines, 'arm: describe' lines, JSON rows) into a
compact table: per run the variant, whether the planner chose the region-fused route, and per arm the temporal kernel / whole sequence in ms.

    python scripts/r04_fmt_arms.py gpurun_out/r04/rf_wide.txt > profiles/...."""
import json
import sys

for path in sys.argv[1:]:
    for ln in open(path):
        ln = ln.strip()
        if ln.startswith("=="):
            print("\n" + ln)
        elif ln.startswith("{"):
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            print(f"   {d['arm']:<52} temporal {d['temporal_ms_med']:8.3f} ms   whole step {d['sequence_ms_med']:8.3f} ms   rest {d['sequence_ms_med'] - d['temporal_ms_med']:7.3f}")
        elif ln.startswith("base:") and "variant=" in ln:
            v = ln.split("variant=")[1].split()[0]
            chunks = ln.split("chunks=")[1].split(" out_slots")[0] if "chunks=" in ln else "?"
            print(f"   [{v}; chunks {chunks}; planner: {'region-fused' if 'region-fused-capable' in ln else 'per-cell route'}]")
        elif "max rel diff" in ln:
            print("      " + ln)
