"""time only the fused PDR+LPMD kernels on config 2 via the engine's HIP-event hooks (no result checks):
python tools/time_tile.py [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metheor_amd
from metheor_amd import synth
from tests import util
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
c = synth.chr19_10m()
eng = metheor_amd.Engine(0)
bt = util.device_batch(c, device="cuda:0")
only = os.environ.get("ONLY", "both")
p = metheor_amd.PdrLpmdParams(want_pdr=only != "lpmd", want_lpmd=only != "pdr")
for _ in range(5):
    eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
eng.sync(); eng.timing_enable(True); eng.timing_reset()
for _ in range(steps):
    eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
t = eng.timing()
print("only", only, {k: round(v[0], 4) for k, v in t.items() if v[1] > 0})
