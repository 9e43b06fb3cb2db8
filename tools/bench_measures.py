#!/usr/bin/env python3
"""Secondary measurements (NOT the BASELINE metric; bench.py is): per-measure pass times on
BASELINE config 2 (S-chr19-10M, device-resident batch) and FDRP/qFDRP on config 4 (50x hotspots).
Prints one JSON object per line.  Usage: python tools/bench_measures.py [--reads N] [--steps K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(eng, fn, steps, warmup=2):
    import torch
    for _ in range(warmup):
        eng.reset(); fn()
    eng.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.reset(); fn()
    eng.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    eng.timing_enable(True); eng.timing_reset()
    for _ in range(3):
        eng.reset(); fn()
    k = {n: round(v[0], 4) for n, v in eng.timing().items() if v[1] > 0}
    eng.timing_enable(False); eng.timing_reset()
    return dt, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=1_000_000)
    ap.add_argument("--hotspot-windows", type=int, default=20000)
    args = ap.parse_args()
    import torch
    import metheor_amd
    from metheor_amd import shard, synth
    from oracle import pyoracle
    from tests import util

    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
    eng = metheor_amd.Engine(0, stream=stream.cuda_stream)
    c = synth.chr19_10m(n_reads=args.reads)
    n = len(c["read_start"])
    bt = util.device_batch(c, device="cuda:0")
    cases = {
        "pdr+lpmd (fused, the BASELINE metric)": lambda: eng.pdr_lpmd_accumulate(bt, metheor_amd.PdrLpmdParams()),
        "me/pm (quartets)": lambda: eng.quartet_accumulate(bt),
        "mhl": lambda: eng.mhl_accumulate(bt),
        "fdrp+qfdrp (-D 40)": lambda: eng.fdrp_accumulate(bt),
        "fdrp+qfdrp (-D 64)": lambda: eng.fdrp_accumulate(bt, max_depth=64),
        "lpmd --pairs table": lambda: eng.lpmd_pairs_accumulate(bt),
    }
    for name, fn in cases.items():
        dt, k = timed(eng, fn, args.steps)
        print(json.dumps({"workload": "S-chr19-10M", "measure": name, "reads": n, "ms_per_pass": round(dt * 1e3, 3),
                          "M_reads_per_s": round(n / dt / 1e6, 1), "kernels_ms": k}), flush=True)
    # CPU oracle on a bounded prefix (1 core), for context only
    ns = min(args.cpu_sample, n)
    sub = shard.slice_region(c, 0, int(c["read_start"][ns - 1]) + 1, halo=0)
    rd = pyoracle.Reads.from_soa(*synth.to_oracle_soa(sub))
    for name, fn in (("pdr", rd.pdr), ("lpmd (incl. pair maps)", rd.lpmd), ("me", rd.me), ("mhl", rd.mhl), ("fdrp", rd.fdrp), ("qfdrp", rd.qfdrp)):
        t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
        print(json.dumps({"workload": "S-chr19-10M prefix", "cpu_oracle": name, "reads": len(rd), "s": round(dt, 3),
                          "M_reads_per_s": round(len(rd) / dt / 1e6, 3), "cores": 1}), flush=True)
    del bt
    # config 4: hotspots at exactly 50x, -D 64 (no sampling)
    h = synth.hotspots(n_windows=args.hotspot_windows)
    hb = util.device_batch(h, device="cuda:0")
    dt, k = timed(eng, lambda: eng.fdrp_accumulate(hb, max_depth=64), args.steps)
    eng.reset(); eng.fdrp_accumulate(hb, max_depth=64); r = eng.fdrp_fetch()
    pairs = float((r["n_reads"].astype(np.float64) * (r["n_reads"] - 1) / 2).sum())
    print(json.dumps({"workload": "S-hotspot-50x (config 4)", "measure": "fdrp+qfdrp -D 64", "reads": len(h["read_start"]),
                      "sites": int(len(r["pos"])), "read_pairs": pairs, "ms_per_pass": round(dt * 1e3, 3),
                      "G_pairs_per_s": round(pairs / dt / 1e9, 3), "kernels_ms": k}), flush=True)
    hs = shard.slice_region(h, 0, int(h["read_start"][min(200_000, len(h["read_start"]) - 1)]) + 1, halo=0)
    hr = pyoracle.Reads.from_soa(*synth.to_oracle_soa(hs))
    t0 = time.perf_counter(); t = hr.fdrp(max_depth=64); dtc = time.perf_counter() - t0
    pc = float((t.cnt[:, 0].astype(np.float64) * (t.cnt[:, 0] - 1) / 2).sum())
    print(json.dumps({"workload": "S-hotspot-50x prefix", "cpu_oracle": "fdrp -D 64", "reads": len(hr), "s": round(dtc, 2),
                      "G_pairs_per_s": round(pc / dtc / 1e9, 5), "cores": 1}), flush=True)
    eng.close()


if __name__ == "__main__":
    main()
