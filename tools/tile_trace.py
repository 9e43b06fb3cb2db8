"""per-tile phase timeline of the tile kernel (library built with MTH_EXTRA_HIPFLAGS=-DMTH_TILE_TRACE): python tools/tile_trace.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["MTH_TILE_TRACE_OUT"] = "/tmp/tile_trace.bin"
import torch, metheor_amd
from metheor_amd import synth, batches
import numpy as np
sparse = len(sys.argv) > 1 and sys.argv[1] == "sparse"
c = synth.make_contig(0, 248_956_422, 16_000_000, 0.0091, np.random.default_rng(3)) if sparse else synth.chr19_10m()
eng = metheor_amd.Engine(0)
bt = batches.device_batch(c, device="cuda:0")
p = metheor_amd.PdrLpmdParams()
for _ in range(4):
    eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
eng.sync()
t = np.fromfile("/tmp/tile_trace.bin", dtype=np.uint64).reshape(-1, 8).astype(np.int64)
print("workload:", "chr1-sized contig at config-3 density" if sparse else "config 2")
names = ["start -> cleared (idx loads issued, clear, tables)", "barrier 1", "lo / hi arrive", "read loop (3 iterations) + LDS atomics",
         "partials + barrier 2", "commit + compaction (barrier 3) + scratch rows + tile_cnt"]
seg = [t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3], t[:, 5] - t[:, 4], t[:, 7] - t[:, 5]]
life = t[:, 7] - t[:, 0]
print("tiles", len(t), "candidates per tile median", int(np.median(t[:, 6])))
print("tile lifetime (ticks of s_memtime, thread 0 of each workgroup): p10 %d  p50 %d  p90 %d" % tuple(np.percentile(life, [10, 50, 90])))
for n, s in zip(names, seg):
    print("  %-60s p50 %7d  mean %7d  (%.0f %% of the mean lifetime)" % (n, np.median(s), s.mean(), 100 * s.mean() / life.mean()))
