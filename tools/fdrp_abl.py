"""temporary: FDRP walk ablation timing"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, metheor_amd
from metheor_amd import synth
from tests import util
c = synth.chr19_10m()
eng = metheor_amd.Engine(0)
eng.timing_enable(True)
bt = util.device_batch(c, device="cuda:0")
for abl in [0, 1, 2, 4, 6, 8, 9]:
    os.environ["MTH_FDRP_ABL"] = str(abl)
    ts = []
    eng.timing_reset()
    for _ in range(4):
        eng.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.fdrp_accumulate(bt); eng.sync(); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
        n = len(eng.fdrp_fetch()["pos"])
    print("abl", abl, "ms", ["%.2f" % t for t in ts], "rows", n, "kernel_ms", {k: round(v[0] / max(v[1], 1), 3) for k, v in eng.timing().items() if v[1]}, flush=True)
