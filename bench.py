#!/usr/bin/env python3
"""bench.py -- M reads/s of the fused PDR+LPMD hot path on MI355X (BASELINE.json metric).

A "step" is one complete pass of the hot path over one resident batch: reset -> linear index ->
tile accumulate (PDR site counters + LPMD pair counts) -> scan -> gather (sorted rows + f32 PDR)
[-> RCCL all-reduce of the 4 LPMD counters when N > 1].  Workload at every N: BASELINE config 2,
"S-chr19-10M" (10 M synthetic 150-bp reads on a 58.6-Mbp contig) PER GPU -- weak scaling, the
contig/region sharding of SURVEY 8(e): rank r owns contig r; per-site rows are disjoint by
construction and only the genome-wide LPMD counters are exchanged.

Usage:  python bench.py --gpus N --steps K --warmup W      (N>1: launched through torch.distributed.run)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (config 2 = 10 M)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="reads of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=20)
    ap.add_argument("--only", choices=["both", "pdr", "lpmd"], default="both",
                    help="experiment knob: time one half of the fused pass (the reported metric needs 'both')")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus or world == 1 and args.gpus == 1, "launch with torch.distributed.run for --gpus > 1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    import metheor_amd
    from metheor_amd import synth
    from tests import util

    # ---- synthetic input, resident in HBM before the timed region --------------------------------
    c = synth.chr19_10m(n_reads=args.reads, seed=1234 + rank)
    c["tid"] = rank
    n_reads, n_calls = len(c["read_start"]), int(c["cpg_off"][-1])
    # a dedicated non-default torch stream: the engine enqueues on it and RCCL orders against it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = metheor_amd.Engine(local_rank, stream=stream.cuda_stream)
    batch = util.device_batch(c, device=dev)
    params = metheor_amd.PdrLpmdParams(want_pdr=args.only != "lpmd", want_lpmd=args.only != "pdr")  # reference CLI defaults
    lp = torch.zeros(4, dtype=torch.int64, device=dev)

    def step():
        eng.reset()
        eng.pdr_lpmd_accumulate(batch, params)
        if world > 1:
            eng.lpmd_export_device(lp.data_ptr())
            dist.all_reduce(lp)                  # RCCL over xGMI: 32 bytes

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # results of the last step (sanity: the job really produced the rows)
    n_sites = eng.pdr_count()
    lg = eng.lpmd_global()
    assert args.only != "both" or (lg["n_read"] == n_reads and n_sites > 0)

    out = None
    if rank == 0:
        total_reads = float(n_reads) * world * args.steps
        value = total_reads / dt / 1e6
        out = {"metric": "M reads/sec (PDR+LPMD, 150bp WGBS)" + ("" if args.only == "both" else " [EXPERIMENT only=%s]" % args.only), "value": round(value, 3), "unit": "M reads/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": "S-chr19-10M (BASELINE config 2): %d x 150bp reads/GPU, chr19 58.6 Mbp, "
                                      "%.2f CpG calls/read, fused PDR+LPMD, reference CLI defaults" % (n_reads, n_calls / n_reads),
                          "reads_per_gpu": n_reads, "cpg_calls_per_gpu": n_calls, "sites_emitted": int(n_sites),
                          "parallelism": "contig-sharded x%d" % world}}

    # ---- roofline leg: dominant kernel timed with HIP events on its own stream ---------------------
    eng.timing_enable(True)
    eng.timing_reset()
    for _ in range(args.roofline_steps):
        eng.reset()
        eng.pdr_lpmd_accumulate(batch, params)
    tm = eng.timing()
    eng.timing_enable(False)
    eng.timing_reset()
    # sites that exist before the min_depth filter are what the 12 B/site output term counts
    eng.reset()
    eng.pdr_lpmd_accumulate(batch, metheor_amd.PdrLpmdParams(min_depth=0, min_cpgs=0))
    n_sites_all = eng.pdr_count()
    if rank == 0:
        dom = "k_pdr_lpmd_tile"
        alg_bytes = 16.0 * n_reads + 5.0 * n_calls + 12.0 * n_sites_all + 32.0
        ms = tm[dom][0]
        achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic = None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pj):
            try:
                t = json.load(open(pj))
                if t.get("reads_per_gpu") == n_reads:
                    traffic = t.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "bytes_per_read": round(alg_bytes / n_reads, 3), "kernel_ms": round(ms, 5),
                           "all_kernels_ms": {k: round(v[0], 5) for k, v in tm.items()}}

    # ---- CPU baseline: the oracle (faithful single-thread port of the reference algorithm) ----------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from metheor_amd import shard
        from oracle import pyoracle
        ns = min(args.cpu_sample, n_reads)
        sub = shard.slice_region(c, 0, int(c["read_start"][ns - 1]) + 1, halo=0)
        rd = pyoracle.Reads.from_soa(*synth.to_oracle_soa(sub))
        # repeat the two passes until >= 10 s of CPU work have been timed (one pass is < 1 s on this host)
        reps, tc = 0, 0.0
        while tc < 10.0 and reps < 200:
            t0 = time.perf_counter()
            rd.pdr()
            rd.lpmd()
            tc += time.perf_counter() - t0
            reps += 1
        out["cpu_baseline"] = {"value": round(len(rd) * reps / tc / 1e6, 4), "unit": "M reads/s", "cores": 1, "kind": "port",
                               "sample": "%d reads of the same workload (pre-decoded SoA), oracle pdr pass then lpmd pass "
                                         "(two passes, as the reference runs them), repeated %d times = %.1f s of CPU work, mean"
                                         % (len(rd), reps, tc)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
