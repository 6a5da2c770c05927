#!/usr/bin/env python3
"""Decode every LZ4 fixture with the kernel AFHIP_LZ4_KERNEL selects and report the first mismatch per fixture with its context."""
import base64, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_blosc_fixtures import recipe
from aggfly_amd import codec
import test_gpu_decode as T
cases = json.load(open(os.path.join(ROOT, "tests", "golden", "blosc_fixtures.json")))["cases"]
nbad = 0
for c in cases:
    chunk = base64.b64decode(c["chunk_b64"]); raw = recipe(c["recipe"], c["n"], c["dtype"], c["seed"])
    got, nerr, host, out_off = T._gpu_decode(torch, [chunk], [raw.nbytes])
    if got[0] is None:
        continue
    r = np.frombuffer(raw.tobytes(), np.uint8)
    d = np.nonzero(got[0] != r)[0]
    if len(d) or nerr:
        nbad += 1
        print(c["cname"], c["shuffle"], c["dtype"], c["recipe"], c["n"], "errors", nerr, "mismatches", len(d), "first", d[:12].tolist())
        if len(d):
            i = int(d[0]); print("  got ", got[0][max(i - 8, 0):i + 8].tolist()); print("  want", r[max(i - 8, 0):i + 8].tolist())
print("fixtures with mismatches:", nbad)
