#!/usr/bin/env python3
"""bench.py -- M reads/s of the fused PDR+LPMD hot path on MI355X (BASELINE.json metric).

A "step" is one complete pass of the hot path over one resident batch: reset -> linear index ->
tile accumulate (PDR site counters + LPMD pair counts) -> gather (sorted rows + f32 PDR, batch totals)
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
    # under torch.distributed.run (RANK set) the RCCL path is used even for a single rank, so that the
    # N > 1 code path can be exercised on a 1-GPU box
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node N for --gpus N > 1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

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
    # the one exchange step of the path: all-reduce(sum) of the 4 LPMD int64 counters (RCCL over xGMI,
    # 32 bytes).  It is issued asynchronously on a ring of buffers so that its latency overlaps the next
    # steps' kernels; a buffer is reused only after its collective has been waited for.
    RING = 4
    lp = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(RING)]
    pending = [None] * RING
    state = {"i": 0}

    def step():
        eng.reset()
        eng.pdr_lpmd_accumulate(batch, params)
        if use_dist:
            k = state["i"] % RING
            state["i"] += 1
            if pending[k] is not None:
                pending[k].wait()                # orders the current stream after that collective; no host block
            eng.lpmd_export_device(lp[k].data_ptr())
            pending[k] = dist.all_reduce(lp[k], async_op=True)

    def fence():
        for k in range(RING):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
        torch.cuda.synchronize()
        if use_dist:
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
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    # the genome-wide LPMD of the last step from the all-reduced counters (what a multi-GPU host reports)
    if use_dist:
        last = lp[(state["i"] - 1) % RING].tolist()
        lpmd_all = float(eng.lpmd_from_counts(last[0], last[1]))
        assert last[2] == n_reads * world or world > 1, (last, n_reads)

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
    if rank == 0 and world == 1 and not use_dist and not args.no_cpu_baseline:
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
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
