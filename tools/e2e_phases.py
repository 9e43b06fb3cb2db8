"""`metheor pdr` end to end on a config-2 BAM written here: wall time of N runs under one or more environments, phases of one.
Usage (GPU box): python tools/e2e_phases.py [reps] [ENV=VAL ...]   (each ENV=VAL is a separate arm next to the default one)"""
import os, statistics, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metheor_amd import hostapi, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 7
arms = [{}] + [dict([a.split("=", 1)]) for a in sys.argv[2:]]
c = synth.chr19_10m()
bam = "/dev/shm/e2e_phases.bam"
hostapi.write_synthetic_bam(bam, c, contig="chr19", seed=7)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metheor_amd", "metheor")
try:
    for rnd in range(2):
        for arm in arms:
            ts = []
            for rep in range(reps):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "pdr", "-i", bam, "-o", "/dev/shm/e2e_phases.tsv"], capture_output=True, text=True, env=dict(os.environ, METHEOR_TIMING="1", **arm))
                ts.append(time.perf_counter() - t0)
                assert r.returncode == 0, r.stderr
            print("%-32s min %.3f median %.3f s  (%.1f / %.1f M reads/s)" % (arm or "default", min(ts), statistics.median(ts), 10 / min(ts), 10 / statistics.median(ts)), flush=True)
            if rnd == 1:
                print("\n".join(l for l in r.stderr.splitlines() if "timing" in l))
finally:
    os.remove(bam)
