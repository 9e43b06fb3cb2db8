#!/usr/bin/env python3
"""bench.py -- M reads/s of the fused PDR+LPMD hot path on MI355X (BASELINE.json metric).

A "step" is one complete pass of the hot path over one resident batch: reset -> linear index ->
tile accumulate (PDR site counters + LPMD pair counts) -> gather (sorted rows + f32 PDR, batch totals).
A job is K steps (batches) and, when N > 1, the path's ONE exchange: the RCCL all-reduce of the 4 genome-wide
LPMD counters (lpmd.rs:51-55 divides them once per run), issued through the C ABI's mth_allreduce_lpmd_rank after
the last step and INSIDE the timed region (--reduce-every-step issues it after every step instead; the line says which,
and carries the collective's own latency either way).  Workload at every N: BASELINE config 2, "S-chr19-10M"
(10 M synthetic 150-bp reads on a 58.6-Mbp contig) PER GPU -- weak scaling, the contig/region sharding of
SURVEY 8(e): rank r owns contig r; per-site rows are disjoint by construction and only the genome-wide LPMD
counters are exchanged.

Usage:  python bench.py --gpus N --steps K --warmup W
N > 1 works both ways: started under torch.distributed.run (RANK/WORLD_SIZE in the environment) this process is one
rank; started plainly it re-launches itself as N ranks (one per GPU, 127.0.0.1 rendezvous) and relays rank 0's line.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 2000 (weak) / 20 (strong)")
    ap.add_argument("--warmup", type=int, default=None, help="default 50 (weak) / 3 (strong)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, the driver's contract): BASELINE config 2 per GPU.  strong: ONE whole-genome WGBS input (BASELINE config 5 "
                         "shape, --strong-reads reads over 24 hg38-sized contigs) region-sharded over the N ranks, one RCCL all-reduce per job")
    ap.add_argument("--strong-reads", type=int, default=1_000_000_000, help="reads of the strong-scaling genome (config 5 = 1 B)")
    ap.add_argument("--plan-only", action="store_true",
                    help="strong mode without a GPU: generate (torch on the CPU), plan, slice and report the partition; everything up to the engine calls")
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU (config 2 = 10 M)")
    ap.add_argument("--cpu-sample", type=int, default=10_000_000, help="reads of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the BAM -> TSV end-to-end leg (N = 1 only)")
    ap.add_argument("--roofline-steps", type=int, default=20)
    ap.add_argument("--soak-seconds", type=float, default=2.0,
                    help="untimed extra passes after the timed region (N = 1) so that an external utilisation sampler sees the GPU busy")
    ap.add_argument("--preheat-seconds", type=float, default=1.0,
                    help="untimed passes BEFORE the warm-up steps (clocks and allocations settle: with the driver's --steps 20 the timed region "
                         "is 2 ms, which an idle GPU spends ramping up); the W warm-up steps and the K timed steps follow unchanged")
    ap.add_argument("--only", choices=["both", "pdr", "lpmd"], default="both",
                    help="experiment knob: time one half of the fused pass (the reported metric needs 'both')")
    ap.add_argument("--share-devices", action="store_true",
                    help="TEST MODE: allow more ranks than GPUs (ranks share devices; RCCL refuses that, so the counters are "
                         "reduced with gloo on the host and the line says \"valid\": false)")
    ap.add_argument("--reduce-every-step", action="store_true",
                    help="N > 1: all-reduce the LPMD counters after every step (default: once per job, after the last step, inside the timed region)")
    ap.add_argument("--no-wgbs", action="store_true", help="skip the WGBS-depth legs (roofline_wgbs, all7, fdrp_pairs; N = 1 only)")
    ap.add_argument("--wgbs-reads", type=int, default=200_000_000, help="reads of the config-3 leg (24 hg38-sized contigs)")
    ap.add_argument("--legs", default="", help="PROFILING: run only these WGBS-depth legs (comma list of roofline_wgbs, all7, fdrp_pairs), print their JSON and exit -- "
                                                "the command tools/profile_round.sh wraps in rocprofv3 for the per-kernel CSVs of those legs")
    ap.add_argument("--no-traffic", action="store_true", help="do not re-measure roofline.traffic with rocprofv3 even when it is on PATH")
    ap.add_argument("--traffic-probe", default="", help="INTERNAL: run a few steps on the arrays saved in this .npz and exit (the process rocprofv3 wraps)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="no GPU work: run only the N-rank scaffolding (spawn, rendezvous, barrier, max over ranks, one JSON line)")
    a = ap.parse_args(argv)
    if a.steps is None:
        a.steps = 20 if a.scaling == "strong" else 2000
    if a.warmup is None:
        a.warmup = 3 if a.scaling == "strong" else 50
    return a


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch(args):
    """this process was started plainly with --gpus N > 1: become the launcher of N ranks"""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        sys.stderr.write(p.stdout)
        return p.returncode or 1
    print(lines[-1], flush=True)
    return 0


def selftest(args, rank, world):
    """the distributed scaffolding alone (tests/test_bench_launcher.py runs it on CPU): same barrier / max-over-ranks
    / rank-0 JSON logic as the real run, a sleep instead of the device pass"""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    per_rank = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(per_rank, dt)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest", "n_gpus": args.gpus, "world_seen": world, "steps": args.steps,
                          "ms_per_step": round(float(dt) / args.steps * 1e3, 4),
                          "per_rank_ms": [round(float(x) / args.steps * 1e3, 4) for x in per_rank]}), flush=True)
    dist.destroy_process_group()
    return 0


def copy_ceiling_gbps(torch, dev, stream):
    """measured device-copy rate on this box (read + write bytes per second of a 1-GiB copy): the practical HBM ceiling"""
    n = 1 << 28
    x = torch.empty(n, dtype=torch.int32, device=dev).fill_(1)
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    reps = 10
    for _ in range(reps):
        y.copy_(x)
    e1.record(stream)
    e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del x, y
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def cpu_info():
    model = ""
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return os.cpu_count(), model


def _phases(stderr):
    """[metheor timing] lines of one run (METHEOR_TIMING=1) -> {phase: seconds}"""
    ph = {}
    for l in stderr.splitlines():
        if l.startswith("[metheor timing]"):
            body = l[len("[metheor timing]"):].rstrip()
            name, sec = body[:-2].rsplit(None, 1) if body.endswith(" s") else (body, "nan")
            try:
                ph[name.strip()] = float(sec)
            except ValueError:
                pass
    return ph


def _run_cli(exe, bam, tsv, reps):
    runs = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "pdr", "-i", bam, "-o", tsv], capture_output=True, text=True, env=dict(os.environ, METHEOR_TIMING="1"))
        dt = time.perf_counter() - t0
        if r.returncode != 0:
            return None, r.stderr[-300:]
        runs.append((dt, _phases(r.stderr)))
    return runs, None


def _split(dt, ph):
    """a run's wall time by phase: load = BGZF inflate + record walk + XM decode on the device (file -> SoA in HBM, host-to-device copy of
    the file included), kernels = the measure's passes + sync, tail = fetch + format + write, startup = everything else (process start, HIP
    runtime start-up -- overlapped with opening the file --, exit)"""
    load = ph.get("device inflate + walk + decode", ph.get("inflate + device record decode", ph.get("host decode (BGZF+BAM+XM)", 0.0)))
    kern = ph.get("H2D + kernels (sync)", 0.0)
    # ("fetch + TSV write" is the CLI's outer phase around "fetch" and "format + write": one or the other, never their sum)
    tail = ph["fetch + TSV write"] if "fetch + TSV write" in ph else ph.get("fetch", 0.0) + ph.get("format + write", 0.0)
    # what the start-up remainder is made of, as far as the process itself can tell: the HIP runtime coming up
    # (hipGetDeviceCount is where it initialises; properties, first stream, first allocations behind it) -- a box effect shows here
    hip = {k[4:]: v for k, v in ph.items() if k.startswith("ctx/")}
    return {"wall_s": round(dt, 4), "startup_s": round(max(dt - load - kern - tail, 0.0), 4), "load_s": round(load, 4), "kernels_s": round(kern, 4),
            "tail_s": round(tail, 4), "device_context_overlapped_s": ph.get("device context (overlapped)", ph.get("device context")),
            "hip_init_s": round(hip.get("hipGetDeviceCount", 0.0), 4), "hip_context_rest_s": round(sum(v for k, v in hip.items() if k != "hipGetDeviceCount"), 4),
            "load_plus_tail_s": round(load + tail, 4)}


def _resident_fraction(path):
    """fraction of the file's pages in the page cache (mincore over a read-only mapping); None when it cannot be told"""
    try:
        import ctypes
        import mmap
        size = os.path.getsize(path)
        if size == 0:
            return None
        libc = ctypes.CDLL(None, use_errno=True)
        ps = mmap.PAGESIZE
        n = (size + ps - 1) // ps
        vec = (ctypes.c_ubyte * n)()
        fd = os.open(path, os.O_RDONLY)
        libc.mmap.restype = ctypes.c_void_p
        libc.mmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_long]
        p = libc.mmap(None, size, 1, 2, fd, 0)                                     # PROT_READ, MAP_PRIVATE
        os.close(fd)
        if p in (None, ctypes.c_void_p(-1).value):
            return None
        libc.mincore.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_ubyte)]
        rc = libc.mincore(ctypes.c_void_p(p), size, vec)
        libc.munmap.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        libc.munmap(ctypes.c_void_p(p), size)
        if rc != 0:
            return None
        return round(sum(1 for b in bytes(vec) if b & 1) / n, 4)
    except Exception:
        return None


def pageable_copy_probe(path, reps=3):
    """one pageable host-to-device copy of a file's bytes out of a fresh read-only mapping whose pages have been touched (what the
    CLI's load phase does with the BAM, chunk by chunk): seconds of the FIRST copy of each fresh mapping (best of reps) -- and of a
    repeated copy of the same mapping, which the runtime serves faster (it is not what a process that reads a file once gets)"""
    import torch
    size = os.path.getsize(path)
    dst = torch.empty(size, dtype=torch.uint8, device="cuda")
    first, again, touch = None, None, None
    for _ in range(reps):
        m = np.memmap(path, dtype=np.uint8, mode="r")
        t0 = time.perf_counter()
        int(np.asarray(m[::4096]).sum())                         # every page mapped (the CLI's block-table scan does this while HIP starts)
        tt = time.perf_counter() - t0
        src = torch.from_numpy(np.asarray(m))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(src)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t0
        t0 = time.perf_counter()
        dst.copy_(src)
        torch.cuda.synchronize()
        d2 = time.perf_counter() - t0
        first = d1 if first is None else min(first, d1)
        again = d2 if again is None else min(again, d2)
        touch = tt if touch is None else min(touch, tt)
        del src, m
    del dst
    return {"file_bytes": size, "touch_pages_s": round(touch, 4), "first_copy_s": round(first, 4), "first_copy_GBps": round(size / first / 1e9, 2),
            "repeated_copy_s": round(again, 4), "repeated_copy_GBps": round(size / again / 1e9, 2)}


def e2e_leg(c, n_reads, large_copies=10):
    """BAM -> TSV: the stand-alone `metheor pdr` executable on a config-2 BAM written here (the north_star's end-to-end
    clause).  Whole-process wall time, file in the page cache, best and median of 5, the median run split by phase; then the
    same on a `large_copies` x larger file (the contig's reads on that many contigs) -- the fixed cost of a process (runtime
    start-up ~0.1 s) is a third of a 10 M-read run and a twentieth of a 100 M-read one."""
    from metheor_amd import hostapi
    d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    bam, tsv = os.path.join(d, "metheor_bench_%d.bam" % os.getpid()), os.path.join(d, "metheor_bench_%d.tsv" % os.getpid())
    exe = os.path.join(ROOT, "metheor_amd", "metheor")
    try:
        t0 = time.perf_counter()
        hostapi.write_synthetic_bam(bam, c, contig="chr19", seed=7)
        t_write = time.perf_counter() - t0
        size = os.path.getsize(bam)
        runs, err = _run_cli(exe, bam, tsv, 6)
        if runs is None:
            return {"error": err}
        runs = sorted(runs[1:], key=lambda x: x[0])
        rows = sum(1 for _ in open(tsv))
        med = runs[len(runs) // 2]
        out = {"what": "`metheor pdr -i <bam> -o <tsv>` whole-process wall time: BGZF inflate + record walk + XM decode + PDR on the "
                       "device, fetch, format, write; BAM in the page cache; 5 runs after one warm-up",
               "reads": n_reads, "bam_bytes": size, "tsv_rows": rows, "best_s": round(runs[0][0], 4), "median_s": round(med[0], 4),
               "M_reads_per_s_best": round(n_reads / runs[0][0] / 1e6, 2), "M_reads_per_s_median": round(n_reads / med[0] / 1e6, 2),
               "median_run": _split(*med), "best_run": _split(*runs[0]),
               "load_phase_M_reads_per_s_median": round(n_reads / max(_split(*med)["load_s"], 1e-9) / 1e6, 1),
               "input_in_page_cache_frac": _resident_fraction(bam), "bam_write_s": round(t_write, 1)}
        # ---- the floor of this load path on THIS box: the file's bytes reach the GPU through one pageable copy out of the
        # page cache -- page-locking the cache's pages is what that copy spends its time on, and it does not parallelise
        # (profiles/r04_h2d_register.md) -- so   wall >= start-up + file bytes / pageable rate + kernels + tail   whatever the kernels do
        try:
            pc = pageable_copy_probe(bam)
            sp = _split(*med)
            floor_s = sp["startup_s"] + pc["first_copy_s"] + sp["kernels_s"] + sp["tail_s"]
            out["floor"] = {"what": "start-up + ONE pageable host-to-device copy of the file out of a FRESH, pre-faulted mapping (as the CLI's is: measured here, best of 3 "
                                    "fresh mappings) + kernels + tail of the median run: the least this load path can take on this box; the load phase also "
                                    "inflates, walks and decodes, overlapped with the copy in pieces",
                            **pc, "floor_s": round(floor_s, 4), "floor_M_reads_per_s": round(n_reads / floor_s / 1e6, 2),
                            "load_over_first_copy": round(sp["load_s"] / pc["first_copy_s"], 3), "needed_for_50M_reads_per_s_s": round(n_reads / 50e6, 4)}
        except Exception as ex:
            out["floor"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        os.remove(bam)
        # ---- the larger file ----
        try:
            import shutil
            free = shutil.disk_usage(d).free
            if large_copies > 1 and free > 3 * large_copies * size:
                t0 = time.perf_counter()
                hostapi.write_synthetic_bam_repeat(bam, c, ["chr19_%d" % k for k in range(large_copies)], seed=7)
                t_w = time.perf_counter() - t0
                big = os.path.getsize(bam)
                runs2, err = _run_cli(exe, bam, tsv, 4)
                if runs2 is None:
                    out["large"] = {"error": err}
                else:
                    runs2 = sorted(runs2[1:], key=lambda x: x[0])
                    nb = n_reads * large_copies
                    m2 = runs2[len(runs2) // 2]
                    out["large"] = {"what": "the same reads on %d contigs in one BAM (%d reads): 3 runs after one warm-up" % (large_copies, nb),
                                    "reads": nb, "bam_bytes": big, "tsv_rows": sum(1 for _ in open(tsv)), "best_s": round(runs2[0][0], 4),
                                    "median_s": round(m2[0], 4), "M_reads_per_s_best": round(nb / runs2[0][0] / 1e6, 2),
                                    "M_reads_per_s_median": round(nb / m2[0] / 1e6, 2), "median_run": _split(*m2),
                                    "load_phase_M_reads_per_s_median": round(nb / max(_split(*m2)["load_s"], 1e-9) / 1e6, 1), "bam_write_s": round(t_w, 1),
                                    "marginal_M_reads_per_s": round((nb - n_reads) / max(m2[0] - med[0], 1e-9) / 1e6, 2)}
            else:
                out["large"] = {"skipped": "not enough room in %s" % d}
        except Exception as ex:
            out["large"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
        return out
    finally:
        for p in (bam, tsv):
            if os.path.exists(p):
                os.remove(p)


def timed_kernels(eng, fn, reps):
    """per-kernel mean ms over reps calls of fn() (HIP events on the engine's stream)"""
    eng.timing_enable(True)
    eng.timing_reset()
    for _ in range(reps):
        eng.reset()
        fn()
    tm = eng.timing()
    eng.timing_enable(False)
    eng.timing_reset()
    return {k: v[0] for k, v in tm.items() if v[1] > 0}


def timed_pass(eng, fn, reps):
    """per-kernel TOTAL ms of one call of fn() (a pass over several batches launches a kernel several times): mean over reps,
    HIP events on the engine's stream; the launches run unpipelined and synchronous where the measure would queue them"""
    eng.timing_enable(True)
    eng.timing_reset()
    for _ in range(reps):
        eng.reset()
        fn()
    eng.sync()
    tm = eng.timing()
    eng.timing_enable(False)
    eng.timing_reset()
    return {k: v[0] * v[1] / reps for k, v in tm.items() if v[1] > 0}


# instruction-issue ceiling of the VALU-bound kernels: 1024 SIMDs x one wave64 vector instruction per ~2.5 cycles for the plain VOP2 forms,
# ~4.3 for compare / select / min-max / three-operand forms (profiles/r02_ubench_valu.md, measured on this part at 2.4 GHz)
SIMDS, CLOCK_GHZ, CYCLES_PER_VALU_FULL, CYCLES_PER_VALU_HALF = 1024, 2.4, 2.5, 4.3


def wgbs_legs(eng, torch, dev, metheor_amd, n_reads_total, legs=("roofline_wgbs", "all7", "fdrp_pairs"), quick=False):
    """The workload the metric is named after (WGBS depth), all device-generated (metheor_amd/synth_device.py):
    roofline_wgbs -- PDR+LPMD on one chr1-sized contig at config-3 density, dominant kernel against 24.5 B/read;
    all7 -- BASELINE config 3, every measure over the 24 contigs queued the way the CLI queues them, one sync;
    fdrp_pairs -- BASELINE config 4 (50x hotspots, -D 64): read pairs per second of the FDRP / qFDRP pass."""
    from metheor_amd import synth, synth_device
    out = {}
    P0 = metheor_amd.PdrLpmdParams()
    gen = torch.Generator(device=dev)
    gen.manual_seed(3)
    if "fdrp_pairs" in legs and "all7" not in legs and "roofline_wgbs" not in legs:
        return fdrp_pairs_leg(eng, torch, dev, metheor_amd, out)
    n1 = int(round(n_reads_total * synth.HG38_LENGTHS[0] / float(sum(synth.HG38_LENGTHS))))
    bt, info = synth_device.make_contig(0, synth.HG38_LENGTHS[0], n1, 0.0091, gen, dev)
    for _ in range(3):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, P0)
    eng.sync()
    tm = timed_kernels(eng, lambda: eng.pdr_lpmd_accumulate(bt, P0), 20)
    eng.reset(); eng.pdr_lpmd_accumulate(bt, metheor_amd.PdrLpmdParams(min_depth=0, min_cpgs=0))
    sites_all = eng.pdr_count()
    dom = max(tm, key=lambda k: tm[k])
    alg = 16.0 * info["n_reads"] + 5.0 * info["n_calls"] + 12.0 * sites_all + 32.0
    step_ms = sum(tm.values())
    out["roofline_wgbs"] = {"workload": "one chr1-sized contig at S-WGBS (config 3) density: %d x 150bp reads on %.1f Mbp, %.2f calls/read, fused PDR+LPMD, CLI defaults"
                                        % (info["n_reads"], info["length"] / 1e6, info["n_calls"] / info["n_reads"]),
                            "bound": "hbm", "kernel": dom, "kernel_ms": round(tm[dom], 5), "algorithmic_bytes_per_launch": alg,
                            "bytes_per_read": round(alg / info["n_reads"], 3), "achieved": round(alg / (tm[dom] * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBPS,
                            "unit": "GB/s", "frac": round(alg / (tm[dom] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                            "whole_step_kernels_ms": round(step_ms, 5), "whole_step_frac": round(alg / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                            "all_kernels_ms": {k: round(v, 5) for k, v in tm.items()}}
    del bt
    if "all7" not in legs:
        return fdrp_pairs_leg(eng, torch, dev, metheor_amd, out) if "fdrp_pairs" in legs else out
    # ---- config 3, all seven measures ----
    t0 = time.perf_counter()
    per_contig, lens, n_tot, c_tot = [], [], 0, 0
    for b, inf in synth_device.wgbs(n_reads=n_reads_total, device=dev):
        per_contig.append(b); lens.append(inf["length"]); n_tot += inf["n_reads"]; c_tot += inf["n_calls"]
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    # What the CLI submits since round 4: the contigs packed into contig GROUPS (include/metheor_hip.h; mth_decoded_group does it to the
    # decoded stream, metheor_amd.batches.group_device_batches to these generated batches) -- 2 batches for the 24 contigs.  The
    # per-contig batches are timed beside them (seven_measures_per_contig_ms: rounds 1-3's figure).
    from metheor_amd import batches as _batches
    resident = _batches.group_device_batches([eng], per_contig, lens)
    # the same roofline figure as roofline_wgbs on the largest batch the CLI actually submits for this workload (the first contig group):
    # a launch 8 x larger spends less of its time filling and draining the chip
    try:
        big = max(resident, key=lambda b: b.n_reads)
        for _ in range(2):
            eng.reset(); eng.pdr_lpmd_accumulate(big, P0)
        eng.sync()
        tmg = timed_kernels(eng, lambda: eng.pdr_lpmd_accumulate(big, P0), 5)
        eng.reset(); eng.pdr_lpmd_accumulate(big, metheor_amd.PdrLpmdParams(min_depth=0, min_cpgs=0))
        sites_g = eng.pdr_count()
        domg = max(tmg, key=lambda k: tmg[k])
        algg = 16.0 * big.n_reads + 5.0 * big.n_cpgs + 12.0 * sites_g + 32.0
        out["roofline_wgbs"]["largest_submitted_batch"] = {
            "workload": "config 3's first contig group: %d reads, %.2f calls/read, one launch" % (big.n_reads, big.n_cpgs / big.n_reads),
            "kernel": domg, "kernel_ms": round(tmg[domg], 5), "algorithmic_bytes_per_launch": algg,
            "achieved": round(algg / (tmg[domg] * 1e-3) / 1e9, 2), "frac": round(algg / (tmg[domg] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
            "whole_step_kernels_ms": round(sum(tmg.values()), 5), "whole_step_frac": round(algg / (sum(tmg.values()) * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
    except Exception as ex:
        out["roofline_wgbs"]["largest_submitted_batch"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    passes = {"pdr+lpmd": lambda b: eng.pdr_lpmd_accumulate(b, P0), "me/pm": lambda b: eng.quartet_accumulate(b),
              "mhl": lambda b: eng.mhl_accumulate(b), "fdrp+qfdrp": lambda b: eng.fdrp_accumulate(b),
              "lpmd --pairs": lambda b: eng.lpmd_pairs_accumulate(b)}
    per, per_c, rows_check = {}, {}, {}
    counts = {"pdr+lpmd": lambda: eng.pdr_count(), "me/pm": lambda: eng.quartet_fetch(min_depth=10)["me"].shape[0], "mhl": lambda: eng.mhl_fetch()["pos"].shape[0],
              "fdrp+qfdrp": lambda: eng.fdrp_fetch()["pos"].shape[0], "lpmd --pairs": lambda: eng.lpmd_pairs_fetch()["pos1"].shape[0]}
    for name, fn in passes.items():
        for which, bs in (((per, resident),) if quick else ((per_c, per_contig), (per, resident))):      # (quick: profiling runs, the grouped form only)
            best = None
            for _ in range(3):
                eng.reset(); eng.sync()
                t0 = time.perf_counter()
                for b in bs:
                    fn(b)
                eng.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            which[name] = best
            rows_check.setdefault(name, []).append(int(counts[name]()))
        if quick:
            rows_check[name].append(rows_check[name][0]); per_c[name] = per[name]
        assert rows_check[name][0] == rows_check[name][1], (name, rows_check[name])      # grouped and per-contig give the same rows
    best_all = None
    for _ in range(3):
        eng.reset(); eng.sync()
        t0 = time.perf_counter()
        for fn in passes.values():
            for b in resident:
                fn(b)
        eng.sync()
        dt = time.perf_counter() - t0
        best_all = dt if best_all is None else min(best_all, dt)
    seven = {k: v for k, v in per.items() if k != "lpmd --pairs"}
    lg = eng.lpmd_global()
    assert lg["n_read"] == n_tot, (lg, n_tot)
    # the same passes over PREPARED batches (mth_batch_prepare, round 5): one read index per batch, built once, shared by the passes
    per_p, prep_s = {}, None
    try:
        eng.sync()
        t0 = time.perf_counter()
        prepared = [eng.batch_prepare(b) for b in resident]
        eng.sync()
        prep_s = time.perf_counter() - t0
        for name, fn in passes.items():
            best = None
            for _ in range(3):
                eng.reset(); eng.sync()
                t0 = time.perf_counter()
                for b in prepared:
                    fn(b)
                eng.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            per_p[name] = best
            assert int(counts[name]()) == rows_check[name][1], (name, "prepared batches give other rows")
        eng.reset(); eng.sync()
        for p_ in prepared:
            p_.release()
    except Exception as ex:
        per_p = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    # what each pass's milliseconds are made of: per-kernel totals over the pass's batches (HIP events; the timed launches run one
    # after the other and every batch synchronously, so their sum sits a little above the queued wall time above)
    kernels_per_pass = {}
    for name, fn in passes.items():
        def whole(fn=fn):
            for b in resident:
                fn(b)
        kernels_per_pass[name] = {k: round(v, 4) for k, v in timed_pass(eng, whole, 2).items()}
    # the four passes of the seven measures side by side: one context (own stream, own work buffers) per pass, the contigs' batches
    # -- shared, read-only -- queued by one host thread per context.  A pass on its own leaves the GPU idle at every kernel boundary and in its
    # latency-bound walks; four of them fill each other's gaps.
    conc = None
    try:
        if quick:
            raise RuntimeError("skipped in a profiling run (--legs)")
        engs = []
        for _ in range(4):
            st = torch.cuda.Stream(device=dev)
            engs.append((metheor_amd.Engine(dev.index, stream=st.cuda_stream), st))
        four = [lambda e, b: e.pdr_lpmd_accumulate(b, P0), lambda e, b: e.quartet_accumulate(b),
                lambda e, b: e.mhl_accumulate(b), lambda e, b: e.fdrp_accumulate(b)]
        order = [3, 2, 1, 0]                       # the longest pass is queued first
        for rep in range(4):
            for e, _ in engs:
                e.reset(); e.sync()
            # one host thread per context (ME / PM syncs once per batch: it must not hold up the other passes' queues)
            import threading
            def run_pass(k):
                for b in per_contig:               # one batch per contig here: four more contexts with group-sized work buffers (16 B a position
                    four[k](engs[k][0], b)         # each, beside the first context's) do not fit 288 GB, and two big batches fill the chip anyway
                engs[k][0].sync()
            th = [threading.Thread(target=run_pass, args=(k,)) for k in order]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            if rep:
                conc = dt if conc is None else min(conc, dt)
        assert engs[0][0].lpmd_global()["n_read"] == n_tot
        rows_c = [engs[0][0].pdr_count(), len(engs[2][0].mhl_fetch()["pos"]), len(engs[3][0].fdrp_fetch()["pos"])]
        for e, _ in engs:
            e.close()
    except Exception as ex:
        conc = None
        out["all7_concurrent_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:200])
    out["all7"] = {"workload": "S-WGBS-200M (BASELINE config 3): %d x 150bp reads over 24 hg38-sized contigs, %.2f calls/read, every contig's batch resident, "
                               "each measure's batches queued back to back (as the CLI queues them); four passes give the seven measures "
                               "(PDR+LPMD fused, ME+PM from one quartet pass, MHL, FDRP+qFDRP from one walk), a fifth the LPMD --pairs table" % (n_tot, c_tot / n_tot),
                   "reads": n_tot, "batches": "%d contig groups for the 24 contigs (%s reads), as the CLI submits them since round 4" % (len(resident), ", ".join(str(b.n_reads) for b in resident)),
                   "per_pass_ms_one_sync_each": {k: round(v * 1e3, 3) for k, v in per.items()},
                   "kernels_ms_per_pass": kernels_per_pass,
                   "kernels_ms_per_pass_what": "per-kernel totals over the pass's batches, HIP events around unpipelined launches (mth_timing_*); the same names "
                                               "appear in profiles/r05_*_all7_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `bench.py --legs all7`)",
                   "roofline_per_pass": {name: {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBPS,
                                                "algorithmic_bytes": (16.0 * n_tot + 5.0 * c_tot) + out_b,
                                                "achieved": round(((16.0 * n_tot + 5.0 * c_tot) + out_b) / per[name] / 1e9, 1),
                                                "frac": round(((16.0 * n_tot + 5.0 * c_tot) + out_b) / per[name] / 1e9 / HBM_PEAK_GBPS, 4)}
                                         for name, out_b in (("pdr+lpmd", 12.0 * rows_check["pdr+lpmd"][1] + 32.0), ("me/pm", 2 * 80.0 * rows_check["me/pm"][1]),
                                                             ("mhl", 8.0 * rows_check["mhl"][1]), ("fdrp+qfdrp", 16.0 * rows_check["fdrp+qfdrp"][1]))},
                   "roofline_per_pass_what": "SURVEY 8(d): every pass reads the input once (16 B/read + 5 B/call) and writes its rows (PDR 12 B/site, ME and PM 80 B/quartet each, MHL 8, FDRP + qFDRP 8 + 8); "
                                             "achieved = those bytes / the pass's wall time above.  The walks of MHL and FDRP are latency / instruction bound: the figure says how far from streaming they are",
                   "seven_measures_ms": round(sum(seven.values()) * 1e3, 3), "all_five_passes_one_sync_ms": round(best_all * 1e3, 3),
                   "prepared_batches": ({"what": "the same passes over batches prepared once with mth_batch_prepare (one read index per batch, shared by the passes; same row counts asserted)",
                                         "prepare_ms_once": round(prep_s * 1e3, 3), "per_pass_ms_one_sync_each": {k: round(v * 1e3, 3) for k, v in per_p.items()},
                                         "seven_measures_ms": round(sum(v for k, v in per_p.items() if k != "lpmd --pairs") * 1e3, 3),
                                         "seven_measures_ms_incl_prepare": round((prep_s + sum(v for k, v in per_p.items() if k != "lpmd --pairs")) * 1e3, 3)}
                                        if "error" not in per_p else per_p),
                   "per_contig_batches": {"what": "the same passes over one batch per contig (24 batches a pass: rounds 1-3's figure); same row counts asserted",
                                          "per_pass_ms_one_sync_each": {k: round(v * 1e3, 3) for k, v in per_c.items()},
                                          "seven_measures_ms": round(sum(v for k, v in per_c.items() if k != "lpmd --pairs") * 1e3, 3)},
                   "rows": {k: v[1] for k, v in rows_check.items()},
                   "seven_measures_concurrent_ms": round(conc * 1e3, 3) if conc else None,
                   "seven_measures_concurrent_what": "the four passes over ONE BATCH PER CONTIG on four contexts of the one GPU (a stream and work buffers each), one host thread per context, wall clock from the first call to the last sync -- the way to fill the chip before the contig groups; compare with per_contig_batches.seven_measures_ms",
                   "G_reads_per_s_seven_concurrent": round(n_tot / conc / 1e9, 3) if conc else None,
                   "G_reads_per_s_seven": round(n_tot / sum(seven.values()) / 1e9, 3),
                   "variant": "unfused: one pass per measure group, each rebuilding the read index",
                   "frac_of_hbm_at_188B_per_read_unfused": round(188.0 * n_tot / sum(seven.values()) / 1e9 / HBM_PEAK_GBPS, 4),
                   "frac_of_hbm_at_39B_per_read_fused": round(39.0 * n_tot / sum(seven.values()) / 1e9 / HBM_PEAK_GBPS, 4),
                   "generate_s": round(t_gen, 2)}
    del resident, per_contig
    torch.cuda.empty_cache()
    if "fdrp_pairs" not in legs:
        return out
    return fdrp_pairs_leg(eng, torch, dev, metheor_amd, out)


def fdrp_pairs_leg(eng, torch, dev, metheor_amd, out):
    """BASELINE config 4 (50x hotspots, -D 64): read pairs per second of the FDRP / qFDRP pass"""
    from metheor_amd import synth_device
    # ---- config 4 ----
    hb, hinf = synth_device.hotspots(device=dev)
    for _ in range(2):
        eng.reset(); eng.fdrp_accumulate(hb, max_depth=64)
    eng.sync()
    best = None
    for _ in range(5):
        eng.reset(); eng.sync()
        t0 = time.perf_counter()
        eng.fdrp_accumulate(hb, max_depth=64)
        eng.sync()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    f = eng.fdrp_fetch()
    nst = f["n_reads"].astype(np.int64)
    pairs = int((nst * (nst - 1) // 2).sum())
    tk = timed_pass(eng, lambda: eng.fdrp_accumulate(hb, max_depth=64), 3)
    # The pass is instruction-issue bound (profiles/r04_fdrp_tile.md, r05 PMC): its roofline is the vector unit.  One evaluation of a
    # (site, read pair) is at least: overlap (and + bcnt of two 64-bit masks = 4), shared calls (4), hamming (xor + and + bcnt = 5),
    # threshold + code (4) -- ~17 plain vector instructions per 64 pairs IF nothing else were issued; the ceiling below prices the
    # pair evaluations alone at the measured full issue rate (every other instruction of the kernel counts against it).
    valu_per_64_pairs_floor = 17.0
    ceiling = SIMDS * CLOCK_GHZ * 1e9 / CYCLES_PER_VALU_FULL / valu_per_64_pairs_floor * 64.0          # site-pairs per second
    out["fdrp_pairs"] = {"workload": "S-hotspot-50x (BASELINE config 4): %d reads in 20000 1-kbp windows at 50x, FDRP + qFDRP, -D 64 (no sampling)" % hinf["n_reads"],
                         "sites": int(len(nst)), "read_pairs": pairs, "pass_ms": round(best * 1e3, 3), "G_pairs_per_s": round(pairs / best / 1e9, 2),
                         "kernels_ms": {k: round(v, 4) for k, v in tk.items()},
                         "kernels_ms_what": "per-kernel totals of one pass (HIP events, mth_timing_*): site discovery = k_build_index + k_pdr_lpmd_tile + k_gather, then "
                                            "k_fdrp_tile (read x read rounds), k_fdrp_chain (ordered f32 sums), k_fdrp_walk (hand-backs), k_fdrp_emit; "
                                            "profiles/r05_*_fdrp_pairs_kernel_stats.csv holds rocprofv3's averages of the same launches",
                         "roofline": {"bound": "valu", "unit": "G site-pairs/s", "achieved": round(pairs / best / 1e9, 2), "peak": round(ceiling / 1e9, 1),
                                      "frac": round(pairs / best / ceiling, 4),
                                      "peak_what": "%d SIMDs x %.1f GHz / %.1f cycles per plain wave64 vector instruction (profiles/r02_ubench_valu.md) / %.0f instructions "
                                                   "per 64 pair evaluations (overlap, shared calls, hamming, code: the arithmetic alone)" % (SIMDS, CLOCK_GHZ, CYCLES_PER_VALU_FULL, valu_per_64_pairs_floor)}}
    return out


def traffic_probe(path):
    """the process rocprofv3 wraps: the arrays of the bench's own batch, a few steps, nothing else"""
    import torch
    import metheor_amd
    from metheor_amd import batches
    z = np.load(path)
    c = {k: z[k] for k in z.files}
    c["tid"] = 0
    c["length"] = int(c.pop("length_"))
    eng = metheor_amd.Engine(0)
    bt = batches.device_batch(c, device="cuda:0")
    p = metheor_amd.PdrLpmdParams()
    for _ in range(8):
        eng.reset(); eng.pdr_lpmd_accumulate(bt, p)
    eng.sync()
    eng.close()
    return 0


def measure_traffic(c, dom):
    """HBM bytes per launch of the dominant kernel from the PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
    WRITE_SIZE in separate --pmc passes (kernel-trace only), FETCH_SIZE doubled (gfx950: it reports half of a wide coalesced
    read).  Returns (bytes, source) or (None, reason)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    d = tempfile.mkdtemp(prefix="metheor_pmc_", dir="/tmp")
    npz = os.path.join(d, "batch.npz")
    np.savez(npz, length_=np.int64(c["length"]), **{k: c[k] for k in ("read_start", "read_end", "read_mapq", "read_fwd", "cpg_off", "cpg_pos", "cpg_rel")})
    env = dict(os.environ, TMPDIR="/tmp")
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            od = os.path.join(d, ctr)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", od, "--", sys.executable, os.path.abspath(__file__),
                                "--traffic-probe", npz], env=env, capture_output=True, text=True, timeout=600)
            acc = []
            for f in glob.glob(od + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr and dom in row["Kernel_Name"]:
                        acc.append(float(row["Counter_Value"]))
            if not acc:
                return None, "rocprofv3 pass %s gave no rows for %s (rc %d)" % (ctr, dom, r.returncode)
            vals[ctr] = sum(acc) / len(acc)
    except Exception as ex:
        return None, "rocprofv3 failed: %s" % str(ex)[:120]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    b = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    return b, ("measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over 8 launches; FETCH_SIZE %.0f KB doubled "
               "(gfx950 correction), WRITE_SIZE %.0f KB" % (vals["FETCH_SIZE"], vals["WRITE_SIZE"]))



def preflight(torch, rank, device_index):
    """one line per rank on stderr: which device this rank drives (a wrong mapping is the first thing to rule out when RCCL fails)"""
    try:
        p = torch.cuda.get_device_properties(device_index)
        uuid = getattr(p, "uuid", "?")
        sys.stderr.write("bench.py preflight: rank %d -> cuda:%d %s uuid %s, %d CUs, %.0f GiB, gcn %s, HSA_ENABLE_IPC_MODE_LEGACY=%s\n"
                         % (rank, device_index, p.name, uuid, p.multi_processor_count, p.total_memory / 2 ** 30,
                            getattr(p, "gcnArchName", "?"), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "unset")))
    except Exception as ex:
        sys.stderr.write("bench.py preflight: rank %d: %s\n" % (rank, ex))


def strong_main(args, rank, world, local_rank, in_rank):
    """--scaling strong: ONE genome (BASELINE config 5 shape), the reads in genome order cut into `world` equal runs
    (metheor_amd.shard.plan_genome: whole contigs, and regions of the contigs a cut falls into), every rank keeps the reads that
    can touch its regions (halo reads re-read, as `metheor --gpus N` does with BGZF blocks), runs the fused PDR+LPMD pass over its
    pieces, and the genome-wide LPMD counters are all-reduced ONCE per job over RCCL (lpmd.rs:51-55)."""
    import torch
    import torch.distributed as dist
    from metheor_amd import shard, synth, synth_device

    use_dist = in_rank
    plan_only = args.plan_only
    if use_dist:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if plan_only:
        dev = torch.device("cpu")
        device_index = -1
    else:
        ndev = torch.cuda.device_count()
        if ndev < 1:
            sys.stderr.write("bench.py: no GPU visible (the engine is HIP-only, there is no CPU fallback)\n")
            return 3
        if world > ndev and not args.share_devices:
            sys.stderr.write("bench.py: --gpus %d needs %d devices, %d visible\n" % (world, world, ndev))
            return 3
        device_index = local_rank % ndev
        torch.cuda.set_device(device_index)
        dev = torch.device("cuda", device_index)
        preflight(torch, rank, device_index)

    lens = synth.HG38_LENGTHS
    tot_len = float(sum(lens))
    contig_reads = [int(round(args.strong_reads * ln / tot_len)) for ln in lens]
    total_reads = sum(contig_reads)
    plan = shard.plan_genome(contig_reads, world)
    mine = {tid: (a, b) for tid, a, b in plan[rank]}

    eng = None
    if not plan_only:
        import metheor_amd
        stream = torch.cuda.Stream(device=dev)
        torch.cuda.set_stream(stream)
        eng = metheor_amd.Engine(device_index, stream=stream.cuda_stream)
        params = metheor_amd.PdrLpmdParams()
    rccl = False
    if eng is not None and use_dist and world <= torch.cuda.device_count():
        import metheor_amd
        ids = [metheor_amd.Engine.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            eng.rccl_init_rank(ids[0], rank, world)
            rccl = True
        except Exception as ex:
            sys.stderr.write("bench.py: rank %d: RCCL communicator could not be created (ncclCommInitRank): %s\n" % (rank, ex))
            return 4

    # every rank generates the same genome (same seed, same generator) contig by contig and keeps its slices
    gen = torch.Generator(device=dev)
    gen.manual_seed(5000)
    pieces, owned, loaded, calls = [], 0, 0, 0
    t0 = time.perf_counter()
    for tid, (ln, n) in enumerate(zip(lens, contig_reads)):
        t, info = synth_device.make_contig_tensors(tid, ln, n, 0.0091, gen, dev)
        if tid in mine:
            a, b = mine[tid]
            beg = 0 if a == 0 else int(t["read_start"][a].item())
            end = ln if b == n else int(t["read_start"][b].item())
            if end > beg:
                sl, n_own = synth_device.slice_region(t, beg, end)
                owned += n_own
                loaded += int(sl["read_start"].shape[0])
                calls += int(sl["cpg_pos"].shape[0])
                pieces.append(sl if plan_only else synth_device.to_batch(sl))
        del t
    t_gen = time.perf_counter() - t0

    def gather_scalar(x, dtype):
        v = torch.tensor([x], dtype=dtype)
        if not use_dist:
            return [v.item()]
        got = [torch.zeros(1, dtype=dtype) for _ in range(world)]
        dist.all_gather(got, v)
        return [g.item() for g in got]

    per_rank_reads = gather_scalar(owned, torch.int64)
    per_rank_loaded = gather_scalar(loaded, torch.int64)
    if plan_only:
        if rank == 0:
            assert sum(per_rank_reads) == total_reads, (per_rank_reads, total_reads)
            print(json.dumps({"metric": "plan only (no GPU work)", "scaling": "strong", "n_gpus": world, "world_seen": world, "reads": total_reads,
                              "per_rank_reads": per_rank_reads, "per_rank_reads_loaded_with_halo": per_rank_loaded,
                              "imbalance_reads": round(max(per_rank_reads) / (sum(per_rank_reads) / world), 4),
                              "per_rank_pieces": gather_scalar(len(pieces), torch.int64)}), flush=True)
        else:
            gather_scalar(len(pieces), torch.int64)
        if use_dist:
            dist.destroy_process_group()
        return 0

    def step(last=False):
        eng.reset()
        for b in pieces:
            eng.pdr_lpmd_accumulate(b, params)
        if rccl and last:
            eng.allreduce_lpmd_rank()

    t_own_done = [0.0]

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        t_own_done[0] = time.perf_counter()      # this rank's work is complete; what follows is the control plane's barrier (gloo, host TCP)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(last=(k == args.warmup - 1))
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(last=(k == args.steps - 1))
    fence()
    dt_mine = time.perf_counter() - t0
    # a rank's own time without the others (no barrier inside): what the imbalance is computed from
    eng.sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        eng.reset()
        for b in pieces:
            eng.pdr_lpmd_accumulate(b, params)
    eng.sync()
    own_ms = (time.perf_counter() - t0) / args.steps * 1e3
    per_rank_dt = gather_scalar(dt_mine, torch.float64)
    per_rank_own = gather_scalar(own_ms, torch.float64)
    lg = eng.lpmd_global()
    sites = gather_scalar(int(eng.pdr_count()), torch.int64)
    if rccl or world == 1:
        assert lg["n_read"] == total_reads, (lg, total_reads)
    elif use_dist:
        # TEST MODE (--share-devices: the ranks share devices, RCCL refuses that): the ranks' counters summed with gloo on host copies --
        # the job's n_read must still be the genome's
        per_rank_n_read = gather_scalar(int(lg["n_read"]), torch.int64)
        assert sum(per_rank_n_read) == total_reads, (per_rank_n_read, total_reads)
    coll = None
    if rccl:
        fence()
        t0 = time.perf_counter()
        for _ in range(10):
            step(last=True)
            eng.lpmd_global()
        t_with = (time.perf_counter() - t0) / 10
        t0 = time.perf_counter()
        for _ in range(10):
            step(last=False)
            eng.lpmd_global()
        t_without = (time.perf_counter() - t0) / 10
        coll = {"serialised_step_with": round(t_with * 1e3, 4), "serialised_step_without": round(t_without * 1e3, 4),
                "all_reduce": round((t_with - t_without) * 1e3, 4)}
    if rank == 0:
        dt = max(per_rank_dt)
        out = {"metric": "M reads/sec (PDR+LPMD, 150bp WGBS)", "value": round(total_reads * args.steps / dt / 1e6, 3), "unit": "M reads/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
               "config": {"workload": "S-WG (BASELINE config 5 shape): ONE genome, %d x 150bp reads over 24 hg38-sized contigs, %.2f calls/read on rank 0, "
                                      "region-sharded over %d rank(s) by equal read runs in genome order (halo reads re-read), fused PDR+LPMD, CLI defaults"
                                      % (total_reads, calls / max(loaded, 1), world),
                          "reads_total": total_reads, "parallelism": "region-sharded x%d" % world},
               "world_seen": world, "devices_visible": torch.cuda.device_count(),
               "collective": ("RCCL ncclAllReduce(int64 x 4, sum) via mth_allreduce_lpmd_rank, world %d, once per job: after the last step, inside the timed region" % world)
                             if rccl else "none",
               "per_rank_reads": per_rank_reads, "per_rank_reads_loaded_with_halo": per_rank_loaded,
               "per_rank_ms_per_step": [round(x / args.steps * 1e3, 4) for x in per_rank_dt],
               "per_rank_own_ms_per_step": [round(x, 4) for x in per_rank_own],
               "imbalance": round(max(per_rank_own) / (sum(per_rank_own) / world), 4),
               "imbalance_reads": round(max(per_rank_reads) / (sum(per_rank_reads) / world), 4),
               "sites_emitted_total": int(sum(sites)), "collective_ms": coll, "generate_s": round(t_gen, 2), "timed_region_s": round(dt, 4)}
        if use_dist and not rccl:
            out["valid"] = False
            out["collective"] = "gloo on host copies (TEST MODE: ranks share devices)"
        print(json.dumps(out), flush=True)
    eng.close()
    if use_dist:
        dist.destroy_process_group()
    return 0


def legs_main(args):
    """`bench.py --legs all7` / `--legs fdrp_pairs` / `--legs roofline_wgbs`: only those legs, one JSON line (what
    tools/profile_round.sh wraps in rocprofv3 so that every reported kernel has its launches in a tracked CSV)"""
    import torch
    import metheor_amd
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    eng = metheor_amd.Engine(0, stream=stream.cuda_stream)
    legs = tuple(x for x in args.legs.split(",") if x)
    out = wgbs_legs(eng, torch, dev, metheor_amd, args.wgbs_reads, legs=legs, quick=True)
    print(json.dumps({"legs": list(legs), **out}), flush=True)
    eng.close()
    return 0


def main():
    args = parse_args()
    if args.traffic_probe:
        return traffic_probe(args.traffic_probe)
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_rank:
        return relaunch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2
    if in_rank:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if args.selftest_launcher:
        return selftest(args, rank, world)
    if args.scaling == "strong":
        return strong_main(args, rank, world, local_rank, in_rank)
    if args.legs:
        return legs_main(args)

    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.stderr.write("bench.py: no GPU visible (the engine is HIP-only, there is no CPU fallback)\n")
        return 3
    shared = world > ndev
    if shared and not args.share_devices:
        sys.stderr.write("bench.py: --gpus %d needs %d devices, %d visible (use --share-devices for a launcher test; its line is not a measurement)\n"
                         % (world, world, ndev))
        return 3
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    use_dist = in_rank
    if use_dist:
        # control plane (barrier, max over ranks, shipping the RCCL id): gloo.  The data-plane collective is the library's.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import metheor_amd
    from metheor_amd import synth
    from metheor_amd import batches

    # ---- synthetic input, resident in HBM before the timed region --------------------------------
    c = synth.chr19_10m(n_reads=args.reads, seed=1234 + rank)
    c["tid"] = rank
    n_reads, n_calls = len(c["read_start"]), int(c["cpg_off"][-1])
    stream = torch.cuda.Stream(device=dev)          # a dedicated non-default stream: the engine enqueues on it
    torch.cuda.set_stream(stream)
    eng = metheor_amd.Engine(device_index, stream=stream.cuda_stream)
    batch = batches.device_batch(c, device=dev)
    params = metheor_amd.PdrLpmdParams(want_pdr=args.only != "lpmd", want_lpmd=args.only != "pdr")  # reference CLI defaults
    collective = "none"
    if use_dist and not shared:
        # the one exchange step of the path: all-reduce(sum) of the 4 LPMD int64 counters, RCCL over xGMI, behind the C ABI
        preflight(torch, rank, device_index)
        ids = [metheor_amd.Engine.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        try:
            eng.rccl_init_rank(ids[0], rank, world)
        except Exception as ex:
            sys.stderr.write("bench.py: rank %d: RCCL communicator could not be created (ncclCommInitRank): %s\n" % (rank, ex))
            return 4
        collective = "RCCL ncclAllReduce(int64 x 4, sum) via mth_allreduce_lpmd_rank, world %d, %s" % (
            world, "after every step" if args.reduce_every_step else "once per job: after the last step, inside the timed region")
    elif use_dist:
        collective = "gloo on host copies (TEST MODE: ranks share devices)"

    rccl = use_dist and not shared and args.only != "pdr"

    def step(last=False):
        eng.reset()
        eng.pdr_lpmd_accumulate(batch, params)
        if rccl and (last or args.reduce_every_step):
            eng.allreduce_lpmd_rank()            # asynchronous: side stream, ordered after this step's kernels

    t_own_done = [0.0]

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        t_own_done[0] = time.perf_counter()      # this rank's work is complete; what follows is the control plane's barrier (gloo, host TCP)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.preheat_seconds > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.preheat_seconds:
            for _ in range(100):
                eng.reset()
                eng.pdr_lpmd_accumulate(batch, params)      # (no collective here: the ranks' loops are not in step)
            eng.sync()
    for k in range(args.warmup):
        step(last=(k == args.warmup - 1))
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(last=(k == args.steps - 1))
    fence()
    dt_mine = time.perf_counter() - t0
    dt_own = t_own_done[0] - t0                  # the same without the closing gloo barrier (reported beside the contract's figure, N > 1)
    dt = dt_mine
    per_rank = [dt_mine]
    dt_before_barrier = dt_own
    if use_dist:
        t = torch.tensor([dt_mine, dt_own], dtype=torch.float64)
        gathered = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, t)
        per_rank = [float(x[0]) for x in gathered]
        dt = max(per_rank)
        dt_before_barrier = max(float(x[1]) for x in gathered)

    # results of the last step (sanity: the job really produced the rows, and the reduce really summed over the ranks)
    n_sites = eng.pdr_count()
    lg = eng.lpmd_global()
    if args.only == "both":
        assert n_sites > 0
        if use_dist and shared:
            t = torch.tensor([lg["n_concordant"], lg["n_discordant"], lg["n_read"], lg["n_valid_read"]], dtype=torch.int64)
            dist.all_reduce(t)
            lg = dict(zip(("n_concordant", "n_discordant", "n_read", "n_valid_read"), t.tolist()))
        assert lg["n_read"] == n_reads * world, (lg, n_reads, world)

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
                          "parallelism": "contig-sharded x%d" % world},
               "preheat_seconds": args.preheat_seconds, "world_seen": world, "devices_visible": ndev, "collective": collective,
               "per_rank_ms_per_step": [round(x / args.steps * 1e3, 4) for x in per_rank],
               "timed_region_s": round(dt, 4)}
        if use_dist:
            # the contract's figure includes the closing barrier of the control plane (gloo over host TCP: 0.1-1 ms against a timed region
            # of K x 0.09 ms); this is the same region up to each rank's own completion, max over ranks -- informational
            out["ms_per_step_before_closing_barrier"] = round(dt_before_barrier / args.steps * 1e3, 4)
        if shared:
            out["valid"] = False

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
        dom = max((k for k in tm if tm[k][1] > 0), key=lambda k: tm[k][0])
        alg_bytes = 16.0 * n_reads + 5.0 * n_calls + 12.0 * n_sites_all + 32.0
        ms = tm[dom][0]
        step_kernels_ms = sum(v[0] for v in tm.values() if v[1] > 0)
        achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic, traffic_src = None, None
        pj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pj):
            try:
                t = json.load(open(pj))
                if t.get("reads_per_gpu") == n_reads and t.get("kernel") == dom:
                    traffic = t.get("hbm_bytes_per_launch")
                    traffic_src = "profiles/pmc_traffic.json: rocprofv3 --pmc passes at commit %s (not re-measured by this run)" % t.get("commit", "?")
            except Exception:
                traffic = None
        ceiling = copy_ceiling_gbps(torch, dev, stream)
        # the same leg over TWO resident batches taken in turn (the second from the device generator, same distributions): one
        # batch's traffic (~250 MB) is the size of the 256-MiB Infinity Cache, two are not -- an HBM statement needs the set to exceed it
        two = None
        if world == 1:
            from metheor_amd import synth_device
            g2 = torch.Generator(device=dev)
            g2.manual_seed(4321)
            batch_b, info_b = synth_device.make_contig(0, synth.CHR19_LEN, n_reads, 0.02, g2, dev)
            for _ in range(2):
                eng.reset(); eng.pdr_lpmd_accumulate(batch_b, params)
            eng.sync()
            eng.timing_enable(True)
            eng.timing_reset()
            for k in range(2 * args.roofline_steps):
                eng.reset()
                eng.pdr_lpmd_accumulate(batch if k % 2 == 0 else batch_b, params)
            tm2 = eng.timing()
            eng.timing_enable(False)
            eng.timing_reset()
            eng.reset(); eng.pdr_lpmd_accumulate(batch_b, metheor_amd.PdrLpmdParams(min_depth=0, min_cpgs=0))
            alg_b = 16.0 * info_b["n_reads"] + 5.0 * info_b["n_calls"] + 12.0 * eng.pdr_count() + 32.0
            alg2 = 0.5 * (alg_bytes + alg_b)
            ms2 = tm2[dom][0]
            two = {"what": "the same kernel timed over two resident batches taken in turn (working set ~2 x %.0f MB > 256 MiB L3)" % (alg_bytes / 1e6),
                   "kernel_ms": round(ms2, 5), "algorithmic_bytes_per_launch_mean": alg2,
                   "achieved": round(alg2 / (ms2 * 1e-3) / 1e9, 2), "frac": round(alg2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)}
            del batch_b
        # `achieved` / `frac` are the TWO-BATCH figures (one resident batch's ~250 MB of traffic sits inside the 256-MiB
        # Infinity Cache, whose hits FETCH_SIZE counts too: not an HBM statement); the single-batch ones stay beside them as
        # `achieved_l3_resident` / `frac_l3_resident`.  N > 1 ranks time one batch only and say so.
        hbm_ach = two["achieved"] if two else round(achieved, 2)
        hbm_frac = two["frac"] if two else round(achieved / HBM_PEAK_GBPS, 5)
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": hbm_ach, "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": hbm_frac,
                           "frac_is": "two resident batches taken in turn (set > 256 MiB Infinity Cache)" if two else "one resident batch (L3-resident)",
                           "achieved_l3_resident": round(achieved, 2), "frac_l3_resident": round(achieved / HBM_PEAK_GBPS, 5),
                           "traffic": traffic, "traffic_source": traffic_src,
                           "copy_ceiling_GBps": round(ceiling, 1), "frac_of_copy_ceiling": round(hbm_ach / ceiling, 5),
                           "algorithmic_bytes_per_launch": alg_bytes,
                           "bytes_per_read": round(alg_bytes / n_reads, 3), "kernel_ms": round(ms, 5),
                           "whole_step_kernels_ms": round(step_kernels_ms, 5),
                           "whole_step_frac": round(alg_bytes / (step_kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5) if step_kernels_ms > 0 else None,
                           "all_kernels_ms": {k: round(v[0], 5) for k, v in tm.items() if v[1] > 0}, "two_batches": two}

    # ---- soak: untimed passes so that an external sampler (rocm-smi every few seconds) can see the GPU working ------
    # ---- the exchange step on its own: latency of one all-reduce of the 32 bytes (every rank takes part) ---------------
    if rccl:
        fence()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.reset()
            eng.pdr_lpmd_accumulate(batch, params)
            eng.allreduce_lpmd_rank()
            eng.lpmd_global()                     # waits for the side stream: step + collective, serialised
        t_with = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            eng.reset()
            eng.pdr_lpmd_accumulate(batch, params)
            eng.lpmd_global()
        t_without = (time.perf_counter() - t0) / 20
        if rank == 0:
            out["collective_ms"] = {"serialised_step_with": round(t_with * 1e3, 4), "serialised_step_without": round(t_without * 1e3, 4),
                                    "all_reduce": round((t_with - t_without) * 1e3, 4)}
    if rank == 0 and world == 1 and args.soak_seconds > 0:
        t0 = time.perf_counter()
        n_soak = 0
        while time.perf_counter() - t0 < args.soak_seconds:
            for _ in range(200):
                step()
            eng.sync()
            n_soak += 200
        out["soak"] = {"seconds": round(time.perf_counter() - t0, 2), "steps": n_soak,
                       "ms_per_step": round((time.perf_counter() - t0) / n_soak * 1e3, 4)}

    # ---- the same steps over TWO contexts of the one GPU (secondary; `value` stays the one-context figure) -----------------------------
    if rank == 0 and world == 1 and not args.no_wgbs:
        try:
            import threading
            st2 = torch.cuda.Stream(device=dev)
            eng2 = metheor_amd.Engine(device_index, stream=st2.cuda_stream)
            k2 = 600

            def run_steps(e):
                for _ in range(k2):
                    e.reset()
                    e.pdr_lpmd_accumulate(batch, params)
                e.sync()
            run_steps(eng2)
            best2 = None
            for _ in range(3):
                th = [threading.Thread(target=run_steps, args=(e,)) for e in (eng, eng2)]
                t0 = time.perf_counter()
                for t in th:
                    t.start()
                for t in th:
                    t.join()
                d2 = time.perf_counter() - t0
                best2 = d2 if best2 is None else min(best2, d2)
            assert eng2.pdr_count() == n_sites
            eng2.close()
            out["two_contexts"] = {"what": "the same steps spread over two contexts (a stream and work buffers each, one host thread each) of the one GPU: the "
                                           "index -> tile -> gather boundaries of one step's kernels are filled by the other's",
                                   "steps": 2 * k2, "ms_per_step": round(best2 / (2 * k2) * 1e3, 4), "M_reads_per_s": round(n_reads * 2 * k2 / best2 / 1e6, 1)}
        except Exception as ex:
            out["two_contexts"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}

    # ---- WGBS depth (the workload the metric is named after): roofline_wgbs, all seven measures on config 3, config-4 pairs/s --
    if rank == 0 and world == 1 and not args.no_wgbs:
        try:
            out.update(wgbs_legs(eng, torch, dev, metheor_amd, args.wgbs_reads))
        except Exception as ex:       # a secondary leg must never cost the bench line
            out["wgbs_error"] = "%s: %s" % (type(ex).__name__, str(ex)[:300])

    # ---- end to end from a BAM file (N = 1): the north_star's 50 M reads/s clause ------------------------------------
    if rank == 0 and world == 1 and not args.no_e2e:
        eng.close()
        eng = None
        del batch
        torch.cuda.empty_cache()
        out["e2e"] = e2e_leg(c, n_reads)

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
        nproc, model = cpu_info()
        out["cpu_baseline"] = {"value": round(len(rd) * reps / tc / 1e6, 4), "unit": "M reads/s", "cores": 1, "kind": "port",
                               "nproc": nproc, "cpu_model": model,
                               "sample": "%d reads of the same workload (pre-decoded SoA), oracle pdr pass then lpmd pass "
                                         "(two passes, as the reference runs them), repeated %d times = %.1f s of CPU work, mean"
                                         % (len(rd), reps, tc)}
        # (i) of SURVEY 8(d): from the BAM file -- single-thread BGZF inflate + record/XM decode (the product's C++ host
        # reader with METHEOR_THREADS=1; the oracle's own BAM loader is pure Python and would only measure Python), then the
        # oracle's two passes, on a 1 M-read BAM of the same generator
        try:
            from metheor_amd import hostapi
            n1 = min(1_000_000, n_reads)
            sub1 = shard.slice_region(c, 0, int(c["read_start"][n1 - 1]) + 1, halo=0)
            d = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
            bam1 = os.path.join(d, "metheor_bench_cpu_%d.bam" % os.getpid())
            hostapi.write_synthetic_bam(bam1, sub1, contig="chr19", seed=7)
            old = os.environ.get("METHEOR_THREADS")
            os.environ["METHEOR_THREADS"] = "1"
            t0 = time.perf_counter()
            soa = hostapi.BamFile(bam1).decode()
            t_dec = time.perf_counter() - t0
            if old is None:
                del os.environ["METHEOR_THREADS"]
            else:
                os.environ["METHEOR_THREADS"] = old
            os.remove(bam1)
            rd1 = pyoracle.Reads.from_soa(soa["tid"], soa["start"], soa["end"], soa["mapq"], soa["fwd"], soa["cpg_off"], soa["cpg_pos"], soa["cpg_rel"])
            t0 = time.perf_counter()
            rd1.pdr()
            rd1.lpmd()
            t_m = time.perf_counter() - t0
            out["cpu_baseline"]["from_bam"] = {"value": round(len(rd1) / (t_dec + t_m) / 1e6, 4), "unit": "M reads/s", "cores": 1,
                                               "decode_s": round(t_dec, 3), "measure_s": round(t_m, 3),
                                               "sample": "%d-read BAM of the same generator: 1-thread zlib inflate + BAM/XM decode (libmetheor_host) + oracle pdr + lpmd" % len(rd1)}
        except Exception as ex:       # the baseline's second figure must never cost the bench line
            out["cpu_baseline"]["from_bam"] = {"error": str(ex)[:200]}
    if rank == 0 and world == 1 and not args.no_traffic and "roofline" in out:
        if eng is not None:
            eng.close()
            eng = None
        b, src = measure_traffic(c, out["roofline"]["kernel"])
        if b is not None:
            out["roofline"]["traffic"], out["roofline"]["traffic_source"] = b, src
        else:
            out["roofline"]["traffic_note"] = src
    if rank == 0:
        print(json.dumps(order_line(out)), flush=True)
    if eng is not None:
        eng.close()
    if use_dist:
        dist.destroy_process_group()
    return 0


def order_line(out):
    """The one JSON line, ordered for a reader that keeps only part of it: the contract's keys first, then every
    secondary figure as a top-level SCALAR, then the long nested objects, and -- because a 2 000-character tail of the line is what a
    driver log keeps -- the same scalars once more in a short `summary` object at the very end."""
    def dig(d, *ks):
        for k in ks:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    sc = {
        "e2e_10M_median": dig(out, "e2e", "M_reads_per_s_median"), "e2e_10M_best": dig(out, "e2e", "M_reads_per_s_best"),
        "e2e_100M_median": dig(out, "e2e", "large", "M_reads_per_s_median"), "e2e_floor_M_reads_per_s": dig(out, "e2e", "floor", "floor_M_reads_per_s"),
        "all7_ms": dig(out, "all7", "seven_measures_ms"), "all7_prepared_ms": dig(out, "all7", "prepared_batches", "seven_measures_ms"),
        "all7_fdrp_pass_ms": dig(out, "all7", "per_pass_ms_one_sync_each", "fdrp+qfdrp"), "all7_mhl_pass_ms": dig(out, "all7", "per_pass_ms_one_sync_each", "mhl"),
        "all7_pdr_pass_ms": dig(out, "all7", "per_pass_ms_one_sync_each", "pdr+lpmd"), "all7_mepm_pass_ms": dig(out, "all7", "per_pass_ms_one_sync_each", "me/pm"),
        "all7_pairs_pass_ms": dig(out, "all7", "per_pass_ms_one_sync_each", "lpmd --pairs"),
        "fdrp_pairs_ms": dig(out, "fdrp_pairs", "pass_ms"),
        "roofline_frac_two_batches": dig(out, "roofline", "frac"), "roofline_frac_l3_resident": dig(out, "roofline", "frac_l3_resident"),
        "roofline_wgbs_frac": dig(out, "roofline_wgbs", "frac"), "roofline_wgbs_frac_group_launch": dig(out, "roofline_wgbs", "largest_submitted_batch", "frac"),
        "cpu_baseline_M_reads_per_s": dig(out, "cpu_baseline", "value"),
    }
    sc = {k: v for k, v in sc.items() if v is not None}
    head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]
    line = {k: out[k] for k in head if k in out}
    line.update(sc)
    for k, v in out.items():
        if k not in line and not isinstance(v, (dict, list)):
            line[k] = v
    for k in ("roofline", "cpu_baseline"):
        if k in out:
            line[k] = out[k]
    for k, v in out.items():
        if k not in line:
            line[k] = v
    line["summary"] = dict(sc, value=out.get("value"), ms_per_step=out.get("ms_per_step"))
    return line


if __name__ == "__main__":
    sys.exit(main())
