"""k_fdrp_walk launch time (HIP events) on BASELINE config 2, device-resident batch: python tools/time_fdrp.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metheor_amd
from metheor_amd import synth
from tests import util
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
c = synth.chr19_10m()
eng = metheor_amd.Engine(0)
bt = util.device_batch(c, device="cuda:0")
for _ in range(2):
    eng.reset(); eng.fdrp_accumulate(bt)
eng.sync()
eng.timing_enable(True); eng.timing_reset()
for _ in range(reps):
    eng.reset(); eng.fdrp_accumulate(bt)
eng.sync()
t = eng.timing()
r = eng.fdrp_fetch()
print("ablate", os.environ.get("METHEOR_FDRP_ABLATE", "0"), {k: round(v[0], 4) for k, v in t.items() if v[1] > 0 and "fdrp" in k},
      "rows", len(r["pos"]), "mean n %.1f" % float(np.mean(r["n_reads"])) if len(r["pos"]) else "")
