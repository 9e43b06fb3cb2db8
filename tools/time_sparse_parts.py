"""the PDR + LPMD pass on one chr1-sized contig at config-3 density, whole and by half (want_pdr / want_lpmd), kernel ms by HIP events:
python tools/time_sparse_parts.py [reads]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, metheor_amd
from metheor_amd import synth, batches
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
c = synth.make_contig(0, 248_956_422, n, 0.0091, np.random.default_rng(3))
eng = metheor_amd.Engine(0)
bt = batches.device_batch(c, device="cuda:0")
for name, kw in (("both", {}), ("pdr only", dict(want_lpmd=False)), ("lpmd only", dict(want_pdr=False)), ("pdr min_cpgs 1", dict(want_lpmd=False, min_cpgs=1))):
    p = metheor_amd.PdrLpmdParams(**kw)
    for _ in range(5):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    eng.sync(); eng.timing_enable(True); eng.timing_reset()
    for _ in range(20):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    t = eng.timing(); eng.timing_enable(False)
    print(name, {k: round(v[0], 4) for k, v in t.items() if v[1] > 0}, flush=True)
