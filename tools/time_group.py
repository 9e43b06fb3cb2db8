"""Per-kernel ms (HIP events, launches unoverlapped) of every measure on config 3's contig groups: python tools/time_group.py [reads] [only]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metheor_amd
from metheor_amd import synth_device, batches
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
eng = metheor_amd.Engine(0)
per, lens = [], []
for b, inf in synth_device.wgbs(n_reads=n_reads, device=dev):
    per.append(b); lens.append(inf["length"])
grp = batches.group_device_batches([eng], per, lens)
del per
P0 = metheor_amd.PdrLpmdParams()
passes = {"pdr+lpmd": lambda b: eng.pdr_lpmd_accumulate(b, P0), "me/pm": lambda b: eng.quartet_accumulate(b), "mhl": lambda b: eng.mhl_accumulate(b),
          "fdrp+qfdrp": lambda b: eng.fdrp_accumulate(b), "pairs": lambda b: eng.lpmd_pairs_accumulate(b)}
for name, fn in passes.items():
    if only and only not in name:
        continue
    for _ in range(2):
        eng.reset()
        for b in grp:
            fn(b)
    eng.sync(); eng.timing_enable(True); eng.timing_reset()
    reps = 3
    for _ in range(reps):
        eng.reset()
        for b in grp:
            fn(b)
    eng.sync()
    t = eng.timing(); eng.timing_enable(False)
    ks = {k: round(v[0] * v[1] / reps, 4) for k, v in t.items() if v[1] > 0}
    print(name, "kernel ms per pass", ks, "sum", round(sum(ks.values()), 3), flush=True)
