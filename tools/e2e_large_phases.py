"""the 100 M-read BAM of bench.py's e2e leg, `metheor pdr` with METHEOR_TIMING=1: every phase line of each run, the wall time, and the
time from the process's last timing line to its exit (python tools/e2e_large_phases.py [copies] [runs])"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metheor_amd import synth, hostapi
copies = int(sys.argv[1]) if len(sys.argv) > 1 else 10
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = synth.chr19_10m()
bam, tsv = "/dev/shm/e2e_large.bam", "/dev/shm/e2e_large.tsv"
t0 = time.perf_counter()
hostapi.write_synthetic_bam_repeat(bam, c, ["chr19_%d" % k for k in range(copies)], seed=7)
print("wrote %.1f GB in %.1f s" % (os.path.getsize(bam) / 1e9, time.perf_counter() - t0), flush=True)
exe = os.path.join(ROOT, "metheor_amd", "metheor")
try:
    for r in range(runs + 1):
        for env in ({}, {"METHEOR_TEARDOWN": "1"}) if r == runs else ({},):
            t0 = time.perf_counter()
            p = subprocess.run([exe, "pdr", "-i", bam, "-o", tsv], capture_output=True, text=True, env=dict(os.environ, METHEOR_TIMING="1", **env))
            dt = time.perf_counter() - t0
            print("run %d %s wall %.3f s rc %d" % (r, env, dt, p.returncode))
            for l in p.stderr.splitlines():
                if "timing" in l:
                    print("   ", l)
finally:
    for f in (bam, tsv):
        if os.path.exists(f):
            os.remove(f)
