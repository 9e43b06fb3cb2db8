#!/usr/bin/env python3
"""one MHL pass (plus a warm-up) on S-chr19 at config-2 depth, for rocprofv3 --pmc runs: python tools/run_mhl_once.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, metheor_amd
from metheor_amd import synth
from tests import util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
c = synth.make_contig(0, int(synth.CHR19_LEN * n / 10_000_000), n, 0.02, np.random.default_rng(1234))
eng = metheor_amd.Engine(0)
bt = util.device_batch(c, device="cuda:0")
for _ in range(3):
    eng.reset(); eng.mhl_accumulate(bt)
eng.sync()
print("rows", len(eng.mhl_fetch()["pos"]))
