"""time k_inflate / k_crc32 / k_decode on a synthetic BAM: python tools/time_inflate.py [reads]  (METHEOR_HIP_LIB selects the build)"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metheor_amd
from metheor_amd import hostapi, synth
from tests.test_gpu_inflate import block_table
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
p = os.path.join(tempfile.gettempdir(), "ti_%d.bam" % n)
if not os.path.exists(p):
    hostapi.write_synthetic_bam(p, synth.make_contig(0, int(58_617_616 * n / 10_000_000), n, 0.02, np.random.default_rng(7)), seed=7)
fb, coff, csize, isize, hbytes, raw = block_table(p)
fbn = np.frombuffer(fb, np.uint8)
eng = metheor_amd.Engine(0)
for _ in range(2):
    eng.bgzf_decode(fbn, coff, csize, isize, hbytes)
eng.timing_enable(True); eng.timing_reset()
for _ in range(4):
    eng.bgzf_decode(fbn, coff, csize, isize, hbytes)
t = eng.timing()
print(os.environ.get("METHEOR_HIP_LIB", "default").split("/")[-1], {k: round(v[0], 3) for k, v in t.items() if v[1] > 0})
