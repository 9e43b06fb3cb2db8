"""one FDRP+qFDRP pass over a config-2-like batch (for rocprofv3 --pmc runs): python tools/run_fdrp_once.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metheor_amd
from metheor_amd import synth
from tests import util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
c = synth.chr19_10m(n_reads=n)
if n < 10_000_000:      # keep the depth of config 2 (25.6x): shrink the contig with the read count
    from metheor_amd import synth as s
    import numpy as np
    L = int(s.CHR19_LEN * n / 10_000_000)
    c = s.make_contig(0, L, n, 0.02, np.random.default_rng(1234))
eng = metheor_amd.Engine(0)
bt = util.device_batch(c, device="cuda:0")
for _ in range(3):
    eng.reset(); eng.fdrp_accumulate(bt)
print("rows", len(eng.fdrp_fetch()["pos"]))
