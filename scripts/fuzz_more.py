import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
import test_gpu_fuzz as f
bad = 0
for seed in range(int(os.environ.get("FUZZ_LO", 28)), int(os.environ.get("FUZZ_HI", 90))):
    if seed % 100 == 0:
        print("seed", seed, "failures so far:", bad, flush=True)      # progress: a silent GPU run is taken to be hung
    try:
        f.test_random_specs_match_oracle(torch, seed)
    except Exception as e:
        bad += 1
        import traceback; print("SEED", seed, "FAILED:", str(e)[:300]); print("".join(traceback.format_exc().splitlines(True)[-14:]))
print("done, failures:", bad, flush=True)
