"""The region-fused fuzz of tests/test_gpu_fuzz.py (`test_random_specs_on_the_region_fused_route`) over further seeds:
FUZZ_LO / FUZZ_HI (default 12 .. 162); seeds with seed % 4 >= 2 carry junction cells (up to five regions per cell)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_fuzz as f
bad = unreached = 0
for seed in range(int(os.environ.get("FUZZ_LO", 12)), int(os.environ.get("FUZZ_HI", 162))):
    if seed % 25 == 0:
        print("seed", seed, "failures so far:", bad, flush=True)
    try:
        f.test_random_specs_on_the_region_fused_route(torch, seed)
    except AssertionError as e:
        if "reached the region-fused route" in str(e):      # numbers matched; none of the seed's plans qualified (e.g. more than six columns)
            unreached += 1
            continue
        bad += 1
        import traceback; print("SEED", seed, "FAILED:", str(e)[:300]); print("".join(traceback.format_exc().splitlines(True)[-10:]))
    except Exception as e:
        bad += 1
        import traceback; print("SEED", seed, "FAILED:", str(e)[:300]); print("".join(traceback.format_exc().splitlines(True)[-10:]))
print("done, failures:", bad, "| seeds whose plans never qualified for the route (numbers matched):", unreached, flush=True)
