"""more seeds of tests/test_gpu_fuzz.py's randomised all-measures scenario than the suite runs (every measure against the oracle, the PDR + LPMD
pass in its dense, streaming and wide forms): python tools/fuzz_more.py [first_seed] [count]"""
import os, sys, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metheor_amd
from tests import test_gpu_fuzz as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
eng = metheor_amd.Engine(0)
bad = []
t0 = time.time()
for seed in range(first, first + count):
    try:
        F.test_random_scenario_all_measures.__wrapped__(eng, seed) if hasattr(F.test_random_scenario_all_measures, "__wrapped__") else F.test_random_scenario_all_measures(eng, seed)
    except Exception as e:
        bad.append(seed)
        print("seed", seed, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
        traceback.print_exc(limit=3)
print("seeds %d..%d: %d failed %s in %.0f s" % (first, first + count - 1, len(bad), bad, time.time() - t0))
sys.exit(1 if bad else 0)
