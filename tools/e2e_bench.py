#!/usr/bin/env python3
"""End-to-end (BAM file -> TSV) timing of the `metheor` executable on a synthetic Bismark BAM.
Not the BASELINE kernel metric: this includes BGZF inflate, BAM parsing, XM decode, H2D, kernels, D2H, TSV.
Usage: python tools/e2e_bench.py [--reads N] [--threads list]"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=1_000_000)
    ap.add_argument("--keep", default=None)
    args = ap.parse_args()
    from metheor_amd import hostapi, synth
    from oracle import bamio
    from tests import util
    n = args.reads
    L = int(synth.CHR19_LEN * n / 10_000_000)          # config-2 depth (25.6x)
    c = synth.make_contig(0, L, n, 0.02, np.random.default_rng(1234))
    d = args.keep or tempfile.mkdtemp()
    bam = os.path.join(d, "syn_%d.bam" % n)
    if not os.path.exists(bam):
        t0 = time.perf_counter()
        hostapi.write_synthetic_bam(bam, c, contig="chr19", seed=1)
        print(json.dumps({"step": "write synthetic BAM (C++ tool, not part of the measurement)", "s": round(time.perf_counter() - t0, 1),
                          "bytes": os.path.getsize(bam)}), flush=True)
    size = os.path.getsize(bam)
    exe = os.path.join(ROOT, "metheor_amd", "metheor")
    for env_threads in (os.environ.get("E2E_THREADS", "default").split(",")):
        env = dict(os.environ)
        if env_threads != "default":
            env["METHEOR_THREADS"] = env_threads
        t0 = time.perf_counter()
        os.environ.update({k: v for k, v in env.items() if k == "METHEOR_THREADS"})
        f = hostapi.BamFile(bam); soa = f.decode(); f.close()
        td = time.perf_counter() - t0
        print(json.dumps({"step": "host decode only (libmetheor_host)", "threads": env_threads, "reads": len(soa["tid"]), "s": round(td, 3),
                          "M_reads_per_s": round(n / td / 1e6, 3), "compressed_MB_per_s": round(size / td / 1e6, 1)}), flush=True)
        for sub in ("pdr", "lpmd", "me", "mhl"):
            out = os.path.join(d, "o_%s.tsv" % sub)
            t0 = time.perf_counter()
            env["METHEOR_TIMING"] = "1"
            r = subprocess.run([exe, sub, "-i", bam, "-o", out], capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr
            if sub == "pdr":
                print(json.dumps({"step": "pdr phases", "threads": env_threads, "stderr": [l for l in r.stderr.splitlines() if "timing" in l]}), flush=True)
            print(json.dumps({"step": "metheor %s end-to-end" % sub, "threads": env_threads, "reads": n, "s": round(dt, 3),
                              "M_reads_per_s": round(n / dt / 1e6, 3), "tsv_lines": sum(1 for _ in open(out))}), flush=True)


if __name__ == "__main__":
    main()
