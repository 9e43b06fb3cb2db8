"""the fused PDR+LPMD kernels on prefixes of config 2 (same density, smaller working sets): is the tile kernel's time per read
a function of where the data lives (L2 32 MiB / L3 256 MiB / HBM)?  python tools/time_tile_sizes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metheor_amd
from metheor_amd import synth, shard, batches
c = synth.chr19_10m()
eng = metheor_amd.Engine(0)
p = metheor_amd.PdrLpmdParams()
for n in (250_000, 500_000, 1_000_000, 2_000_000, 5_000_000, 10_000_000):
    end = int(c["read_start"][n - 1]) + 1 if n < len(c["read_start"]) else c["length"]
    sub = shard.slice_region(c, 0, end, halo=0)
    sub["length"] = end
    bt = batches.device_batch(sub, region=(0, end), device="cuda:0")
    for _ in range(5):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    eng.sync(); eng.timing_enable(True); eng.timing_reset()
    for _ in range(100):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    t = eng.timing(); eng.timing_enable(False); eng.timing_reset()
    k = t["k_pdr_lpmd_tile"][0]
    nr = len(sub["read_start"]); nc = int(sub["cpg_off"][-1])
    print("reads %9d  input %6.1f MB  tile kernel %.4f ms  %.3f ns/read  (%.0f GB/s of the 9 B/read + 5 B/call it reads)" % (nr, (9 * nr + 5 * nc) / 1e6, k, k * 1e6 / nr, (9 * nr + 5 * nc) / k / 1e6))
