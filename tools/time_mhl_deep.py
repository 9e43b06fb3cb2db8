import os, sys, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tools")
import numpy as np, metheor_amd
from metheor_amd import synth
from tests import util
from bench_measures import timed
eng = metheor_amd.Engine(0)
for name, L, n in (("100x", 3_000_000, 2_000_000), ("3000x amplicons", 40_000, 800_000)):
    c = synth.make_contig(0, L, n, 0.02, np.random.default_rng(5))
    bt = util.device_batch(c, device="cuda:0")
    dt, k = timed(eng, lambda: eng.mhl_accumulate(bt), 3)
    print(json.dumps({"case": name, "ms": round(dt * 1e3, 3), "k": k}))
