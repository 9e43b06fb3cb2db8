#!/usr/bin/env python3
"""`metheor <sub>` on one BAM as 1 process and as N shards (python -m metheor_amd.sharded), wall time min / median.
On a 1-GPU box the shards share the GPU: what this shows there is the overlap of the host-side phases (context creation,
file reads, TSV formatting) -- the scaling over GPUs is the driver's to measure.
Usage: python tools/e2e_sharded.py <bam> <sub> <reps> <N> [<N> ...]"""
import os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metheor_amd import sharded

bam, sub, reps = sys.argv[1], sys.argv[2], int(sys.argv[3])
exe = os.path.join(ROOT, "metheor_amd", "metheor")
ref = None
for n in [int(x) for x in sys.argv[4:]]:
    ts = []
    out = "/tmp/e2e_sharded_%d.tsv" % n
    for _ in range(reps):
        t0 = time.perf_counter()
        if n == 1:
            r = subprocess.run([exe, sub, "-i", bam, "-o", out], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
        else:
            assert sharded.run(n, [sub, "-i", bam, "-o", out], gpus=1) == 0
        ts.append(time.perf_counter() - t0)
    data = open(out, "rb").read()
    if ref is None:
        ref = data
    print("%d shard(s): min %.3f  median %.3f s   output %s" % (n, min(ts), statistics.median(ts),
          "identical" if data == ref else "DIFFERENT"), flush=True)
