"""more seeds of tests/test_gpu_fileorder.py's randomised unsorted-file scenario (PDR / MHL / FDRP / qFDRP through the CLI against the oracle in
file order): python tools/fuzz_fileorder.py [first_seed] [count]"""
import os, sys, tempfile, time, traceback, pathlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_fileorder as F
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = []
t0 = time.time()
for seed in range(first, first + count):
    with tempfile.TemporaryDirectory() as d:
        try:
            F.test_random_unsorted_files(pathlib.Path(d), seed)
        except Exception as e:
            bad.append(seed)
            print("seed", seed, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
            traceback.print_exc(limit=3)
print("seeds %d..%d: %d failed %s in %.0f s" % (first, first + count - 1, len(bad), bad, time.time() - t0))
sys.exit(1 if bad else 0)
