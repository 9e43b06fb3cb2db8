"""per-workgroup timeline of k_pdr_lpmd_runs from a -DMTH_RUNS_TRACE build (tools/mkab.sh trace -DMTH_RUNS_TRACE):
METHEOR_HIP_LIB=$PWD/abx/libtrace.so python tools/runs_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metheor_amd
from metheor_amd import synth
from tests import util
c = synth.chr19_10m()
eng = metheor_amd.Engine(0)
bt = util.device_batch(c, device="cuda:0")
p = metheor_amd.PdrLpmdParams()
for _ in range(5):
    eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
eng.sync()
out = "/tmp/runs_trace.bin"
os.environ["MTH_RUNS_TRACE_OUT"] = out
os.environ["MTH_PIPELINE"] = "0"
eng.reset(); eng.pdr_lpmd_accumulate(bt, p); eng.sync()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4, 8)
ntile = (t[:, :, 7] & np.uint64(0xffffffff)).astype(np.float64); iters = (t[:, :, 7] >> np.uint64(32)).astype(np.float64)
t = t.astype(np.float64)
life = t[:, :, 1]
print("workgroups", t.shape[0], "life kticks: mean %.0f p5 %.0f p95 %.0f max %.0f" % (life.mean() / 1e3, np.percentile(life, 5) / 1e3, np.percentile(life, 95) / 1e3, life.max() / 1e3))
for k, nm in [(2, "preamble"), (3, "read loops"), (0, " of it: load wait"), (4, "barrier1 wait"), (5, "compaction"), (6, "barrier2 wait")]:
    v = t[:, :, k]
    print("%-18s %5.1f %% of life; per tile %.0f ticks; per iteration %.0f" % (nm, 100 * v.sum() / life.sum(), (v.sum() / ntile.sum()), v.sum() / max(iters.sum(), 1)))
print("tiles per workgroup %.2f, iterations per wave and tile %.2f" % (ntile[:, 0].mean(), iters.sum() / ntile.sum()))
