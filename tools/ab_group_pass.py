"""wall-clock ms of one measure's pass over config 3's contig groups under the current environment: python tools/ab_group_pass.py mhl|fdrp|pdr|me|pairs [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metheor_amd
from metheor_amd import synth_device, batches
which = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0"); eng = metheor_amd.Engine(0)
per, lens = [], []
for b, inf in synth_device.wgbs(n_reads=200_000_000, device=dev):
    per.append(b); lens.append(inf["length"])
grp = batches.group_device_batches([eng], per, lens); del per
P0 = metheor_amd.PdrLpmdParams()
fn = {"pdr": lambda b: eng.pdr_lpmd_accumulate(b, P0), "me": lambda b: eng.quartet_accumulate(b), "mhl": lambda b: eng.mhl_accumulate(b),
      "fdrp": lambda b: eng.fdrp_accumulate(b), "pairs": lambda b: eng.lpmd_pairs_accumulate(b)}[which]
ts = []
for r in range(reps + 2):
    eng.reset(); eng.sync()
    t0 = time.perf_counter()
    for b in grp: fn(b)
    eng.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
print(which, {k: os.environ[k] for k in os.environ if k.startswith("MTH_") or k.startswith("METHEOR_")}, "best %.3f median %.3f" % (min(ts[2:]), sorted(ts[2:])[len(ts[2:]) // 2]))
