#!/usr/bin/env python3
"""repeat `metheor <sub> -i <bam>` under several environments and report min / median wall time (box noise is +-20 %):
python tools/e2e_repeat.py <bam> <sub> <reps> NAME=VAL[,NAME=VAL] ..."""
import os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bam, sub, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
exe = os.environ.get("METHEOR_EXE") or os.path.join(ROOT, "metheor_amd", "metheor")
for spec in sys.argv[4:]:
    env = dict(os.environ)
    if spec != "-":
        for kv in spec.split(","):
            k, v = kv.split("="); env[k] = v
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run([exe, sub, "-i", bam, "-o", "/tmp/e2e_repeat.tsv"], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        ts.append(time.perf_counter() - t0)
    print("%-40s min %.3f  median %.3f  max %.3f s" % (spec, min(ts), statistics.median(ts), max(ts)), flush=True)
