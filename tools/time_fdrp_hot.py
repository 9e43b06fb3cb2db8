"""k_fdrp_walk on BASELINE config 4 (50x hotspots, -D 64), HIP events: python tools/time_fdrp_hot.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metheor_amd
from metheor_amd import synth_device
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
hb, hinf = synth_device.hotspots(device="cuda:0")
eng = metheor_amd.Engine(0)
for _ in range(2):
    eng.reset(); eng.fdrp_accumulate(hb, max_depth=64)
eng.sync()
eng.timing_enable(True); eng.timing_reset()
for _ in range(reps):
    eng.reset(); eng.fdrp_accumulate(hb, max_depth=64)
eng.sync()
t = eng.timing()
r = eng.fdrp_fetch()
n = r["n_reads"].astype(np.int64)
print("ablate", os.environ.get("METHEOR_FDRP_ABLATE", "0"), {k: round(v[0], 4) for k, v in t.items() if v[1] > 0 and "fdrp" in k},
      "rows", len(n), "pairs %.3e" % float((n * (n - 1) // 2).sum()), "mean n %.1f" % float(n.mean()),
      "checksum %.6f %.6f" % (float(r["fdrp"].astype(np.float64).sum()), float(np.nansum(r["qfdrp"].astype(np.float64)))))
print("all kernels", {k: round(v[0] / reps, 4) for k, v in t.items() if v[1] > 0})
