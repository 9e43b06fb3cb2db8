import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import metheor_amd
from metheor_amd import synth
from oracle import pyoracle
from tests import util
c = synth.make_contig(2, 1_000_000, 200_000, 0.05, np.random.default_rng(31))
reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
eng = metheor_amd.Engine(0)
eng.quartet_accumulate(util.device_batch(c), min_qual=10)
d = eng.quartet_fetch(0)
op = reads.pm(min_depth=0, min_qual=10)
order = np.lexsort((d["pos"][:,3], d["pos"][:,2], d["pos"][:,1], d["pos"][:,0], d["tid"]))
dp, cnt = d["pm"][order], d["cnt"][order]
bad = np.nonzero(dp.view(np.uint32) != op.val.view(np.uint32))[0]
print("rows", len(dp), "mismatching", len(bad), "max abs diff", np.abs(dp - op.val).max())
f = np.float32
def pm_np(c):
    t = f(c.sum()); pm = f(1.0)
    for x in c:
        p = f(x) / t
        pm = f(pm - f(p * p))
    return pm
def pm_np_fma(c):   # what an FMA-contracted device would compute: pm = fma(-p, p, pm)
    t = f(c.sum()); pm = f(1.0)
    for x in c:
        p = f(x) / t
        pm = f(np.float64(pm) - np.float64(p) * np.float64(p))
    return pm
agree_or = agree_dev = agree_fma = 0
for i in bad[:2000]:
    r = pm_np(cnt[i]); agree_or += r.view(np.uint32) == op.val[i].view(np.uint32); agree_dev += r.view(np.uint32) == dp[i].view(np.uint32)
    agree_fma += pm_np_fma(cnt[i]).view(np.uint32) == dp[i].view(np.uint32)
print("of %d mismatches: numpy-sequential == oracle: %d ; == device: %d ; fma-model == device: %d" % (min(len(bad),2000), agree_or, agree_dev, agree_fma))
for i in bad[:3]:
    print(cnt[i].tolist(), "dev %r oracle %r numpy %r" % (dp[i], op.val[i], pm_np(cnt[i])))
