"""How much of the tile kernel's time comes from reads with more than 8 calls (the memory-loop tails)?
Config 2 as it is vs the same batch with every read's calls capped at `cap` (default 8).  Usage: python tools/tile_tail_probe.py [cap] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metheor_amd
from metheor_amd import synth
from tests import util


def cap_calls(c, cap):
    off = c["cpg_off"].astype(np.int64)
    n = np.diff(off)
    k = np.arange(len(c["cpg_pos"]), dtype=np.int64) - np.repeat(off[:-1], n)
    keep = k < cap
    d = dict(c)
    no = np.zeros(len(n) + 1, np.int64)
    np.cumsum(np.minimum(n, cap), out=no[1:])
    d["cpg_off"] = no.astype(np.uint32)
    d["cpg_pos"] = c["cpg_pos"][keep]
    d["cpg_rel"] = c["cpg_rel"][keep]
    return d, int((n > cap).sum())


def time_it(eng, bt, reps, **kw):
    p = metheor_amd.PdrLpmdParams(**kw)
    for _ in range(5):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(reps):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    t = eng.timing(); eng.timing_enable(False); eng.timing_reset()
    return {k: round(v[0], 5) for k, v in t.items() if v[1]}


cap = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = synth.chr19_10m()
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
eng = metheor_amd.Engine(0, stream=stream.cuda_stream)
b0 = util.device_batch(c, device="cuda:0")
c2, n_long = cap_calls(c, cap)
b1 = util.device_batch(c2, device="cuda:0")
for rnd in range(2):
    for kw in ({}, {"want_lpmd": False}, {"want_pdr": False}):
        print(kw or "fused", "as generated", time_it(eng, b0, reps, **kw)["k_pdr_lpmd_tile"], " capped at %d (%.2f %% of reads affected)" % (cap, 100.0 * n_long / len(c["read_start"])),
              time_it(eng, b1, reps, **kw)["k_pdr_lpmd_tile"])
