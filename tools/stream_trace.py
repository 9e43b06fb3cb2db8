"""per-wave timeline of the streaming kernel (library built with -DMTH_STREAM_TRACE): python tools/stream_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ["MTH_STREAM"] = "1"
os.environ["MTH_STREAM_TRACE_OUT"] = "/tmp/stream_trace.bin"
import torch, metheor_amd
from metheor_amd import synth
from tests import util
c = synth.chr19_10m()
eng = metheor_amd.Engine(0)
bt = util.device_batch(c, device="cuda:0")
p = metheor_amd.PdrLpmdParams()
for _ in range(3):
    eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
eng.sync()
t = np.fromfile("/tmp/stream_trace.bin", dtype=np.uint64).reshape(-1, 4)
t0 = t[:, 0].min()
beg, end = (t[:, 0] - t0).astype(np.int64), (t[:, 1] - t0).astype(np.int64)
dur = end - beg
xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64)
hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
halo = (t[:, 3] >> np.uint64(32)).astype(np.int64)
print("waves", len(t), "kernel span", end.max(), "ticks")
print("begin  pct 0/50/90/99/100", np.percentile(beg, [0, 50, 90, 99, 100]).astype(int))
print("end    pct 0/1/10/50/90/100", np.percentile(end, [0, 1, 10, 50, 90, 100]).astype(int))
print("dur    pct 0/10/50/90/100", np.percentile(dur, [0, 10, 50, 90, 100]).astype(int), "mean", int(dur.mean()))
print("halo   pct 50/99/100", np.percentile(halo, [50, 99, 100]).astype(int))
late = beg > np.percentile(dur, 10) * 0.5
print("waves that begin after half of a short wave's duration:", int(late.sum()))
key = xcc * 1000 + se * 100 + sh * 16 + cu
u, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu):", len(u), "waves per cu min/median/max", cnt.min(), int(np.median(cnt)), cnt.max())
for q in (0, 1, 2, 3):
    m = (np.arange(len(t)) % 4) == q
    print(" wave-in-wg", q, "mean dur", int(dur[m].mean()))
# per-XCD view (the tick counters of different XCDs are not aligned: times relative to the XCD's first begin)
for x in range(8):
    m = xcc == x
    if not m.any(): continue
    b = beg[m] - beg[m].min(); e_ = end[m] - beg[m].min()
    print("xcc", x, "waves", int(m.sum()), "begin p50/p100", int(np.percentile(b, 50)), int(b.max()), "end p10/p50/p100", int(np.percentile(e_, 10)), int(np.percentile(e_, 50)), int(e_.max()), "dur mean", int(dur[m].mean()))
# per-CU mean duration spread inside XCD 0
m0 = xcc == 0
ks = key[m0]
for k in np.unique(ks)[:8]:
    mm = m0 & (key == k)
    b = beg[mm] - beg[m0].min()
    print(" cu", k, "n", int(mm.sum()), "dur min/mean/max", dur[mm].min(), int(dur[mm].mean()), dur[mm].max(), "begin min/max", b.min(), b.max())
# duration against wave index (position in the genome)
idx = np.arange(len(t))
for lo in range(0, len(t), 1000):
    print(" waves", lo, "dur mean", int(dur[lo:lo + 1000].mean()))
simd = (hw >> 4) & 3; wave_id = hw & 0xf
for sd in range(4):
    print(" simd", sd, "dur mean", int(dur[simd == sd].mean()), "n", int((simd == sd).sum()))
