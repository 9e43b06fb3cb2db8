"""Interleaved A/B of `metheor pdr` end to end on a config-2 BAM: whole-process wall time and the load phase of every run.
Usage (GPU box): python tools/e2e_ab.py [reps] [ENV=VAL[,ENV=VAL] ...]   (each argument is one arm next to the default one)"""
import os, re, statistics, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metheor_amd import hostapi, synth
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
arms = [{}] + [dict(kv.split("=", 1) for kv in a.split(",")) for a in sys.argv[2:]]
c = synth.chr19_10m()
bam = "/dev/shm/e2e_ab.bam"
hostapi.write_synthetic_bam(bam, c, contig="chr19", seed=7)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metheor_amd", "metheor")
wall = [[] for _ in arms]
load = [[] for _ in arms]
ctxt = [[] for _ in arms]
fetch = [[] for _ in arms]
fmtw = [[] for _ in arms]
try:
    for rep in range(reps + 1):
        for k, arm in enumerate(arms):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "pdr", "-i", bam, "-o", "/dev/shm/e2e_ab.tsv"], capture_output=True, text=True, env=dict(os.environ, METHEOR_TIMING="1", **arm))
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            if rep == 0:
                continue                      # warm-up round
            wall[k].append(dt)
            m = re.search(r"device inflate \+ walk \+ decode\s+([0-9.]+) s", r.stderr)
            load[k].append(float(m.group(1)) if m else float("nan"))
            m = re.search(r"device context \(overlapped\)\s+([0-9.]+) s", r.stderr)
            ctxt[k].append(float(m.group(1)) if m else float("nan"))
            m = re.search(r"fetch \+ TSV write\s+([0-9.]+) s", r.stderr)
            fetch[k].append(float(m.group(1)) if m else float("nan"))
            m = re.search(r"  format \+ write\s+([0-9.]+) s", r.stderr)
            fmtw[k].append(float(m.group(1)) if m else float("nan"))
    for k, arm in enumerate(arms):
        w, l, x = sorted(wall[k]), sorted(load[k]), sorted(ctxt[k])
        q = lambda v, f: v[int(f * (len(v) - 1))]
        print("%-44s wall min %.3f q25 %.3f med %.3f | load min %.3f q25 %.3f med %.3f | ctx min %.3f med %.3f | fetch+write med %.3f (write %.3f) | best %.1f M reads/s" %
              (arm or "default", w[0], q(w, .25), q(w, .5), l[0], q(l, .25), q(l, .5), x[0], q(x, .5), q(sorted(fetch[k]), .5), q(sorted(fmtw[k]), .5), 10 / w[0]), flush=True)
finally:
    os.remove(bam)
