#!/usr/bin/env python3
"""Every measure on ONE sparse WGBS-like contig (config-3 density): ms per pass and per kernel.
Usage: python tools/time_sparse.py [--len 248956422] [--reads 16000000] [--density 0.0091]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--len", type=int, default=248_956_422)
    ap.add_argument("--reads", type=int, default=16_000_000)
    ap.add_argument("--density", type=float, default=0.0091)
    ap.add_argument("--only", default="", help="substring of the measure names to run")
    args = ap.parse_args()
    import metheor_amd
    from metheor_amd import synth
    from tests import util
    from bench_measures import timed
    eng = metheor_amd.Engine(0)
    c = synth.make_contig(0, args.len, args.reads, args.density, np.random.default_rng(3))
    bt = util.device_batch(c, device="cuda:0")
    for name, fn in (("me/pm", lambda: eng.quartet_accumulate(bt)), ("pairs", lambda: eng.lpmd_pairs_accumulate(bt)),
                     ("pdr+lpmd", lambda: eng.pdr_lpmd_accumulate(bt, metheor_amd.PdrLpmdParams())),
                     ("mhl", lambda: eng.mhl_accumulate(bt)), ("fdrp+qfdrp", lambda: eng.fdrp_accumulate(bt))):
        if args.only and args.only not in name:
            continue
        dt, k = timed(eng, fn, 20 if args.only else 5)
        print(json.dumps({"measure": name, "reads": args.reads, "ms_per_pass": round(dt * 1e3, 3), "kernels_ms": k}), flush=True)


if __name__ == "__main__":
    main()
