"""write a synthetic BAM and print the decoder's phase timings: python tools/decode_phases.py [reads] [threads]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metheor_amd import hostapi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
os.environ["METHEOR_TIMING"] = "1"
c = synth.make_contig(0, int(synth.CHR19_LEN * n / 10_000_000), n, 0.02, np.random.default_rng(1234))
p = os.path.join(tempfile.mkdtemp(), "x.bam")
hostapi.write_synthetic_bam(p, c, seed=1)
for th in (sys.argv[2:] or ["64"]):
    os.environ["METHEOR_THREADS"] = th
    for rep in range(2):
        t0 = time.perf_counter(); f = hostapi.BamFile(p); f.decode(); dt = time.perf_counter() - t0; f.close()
        print("threads", th, "rep", rep, "total %.3f s" % dt, flush=True)
