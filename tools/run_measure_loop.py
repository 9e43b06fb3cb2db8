"""one measure, a few passes over the resident config-2 batch (the process the PMC passes wrap): python tools/run_measure_loop.py pairs|mhl|quartet|fdrp [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metheor_amd
from metheor_amd import synth, batches
what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
c = synth.chr19_10m()
eng = metheor_amd.Engine(0)
bt = batches.device_batch(c, device="cuda:0")
fn = {"pairs": lambda: eng.lpmd_pairs_accumulate(bt), "mhl": lambda: eng.mhl_accumulate(bt), "quartet": lambda: eng.quartet_accumulate(bt),
      "fdrp": lambda: eng.fdrp_accumulate(bt)}[what]
for _ in range(reps):
    eng.reset(); fn()
eng.sync()
eng.timing_enable(True); eng.timing_reset()
for _ in range(3):
    eng.reset(); fn()
print({k: round(v[0], 4) for k, v in eng.timing().items() if v[1] > 0})
