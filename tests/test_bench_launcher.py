"""bench.py's N-rank scaffolding on CPU: started plainly with --gpus 2 it must launch two ranks
itself (torch.distributed.run, 127.0.0.1 rendezvous), run barrier + max-over-ranks, and print ONE JSON line from rank 0;
started under torch.distributed.run it must be a rank.  --selftest-launcher replaces the device pass by a sleep, so this
runs without a GPU; the same code path carries the real run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _one_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_plain_invocation_spawns_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "1", "--selftest-launcher"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _one_line(r.stdout)
    assert j["n_gpus"] == 2 and j["world_seen"] == 2 and j["steps"] == 5 and len(j["per_rank_ms"]) == 2
    assert j["ms_per_step"] == max(j["per_rank_ms"])          # the max over ranks is what is reported


def test_under_torchrun_is_a_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest-launcher"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _one_line(r.stdout)["world_seen"] == 2


def test_world_mismatch_is_refused():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-launcher"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


def test_single_gpu_without_device_fails_loudly():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, BENCH, "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr


# ---- --scaling strong: ONE genome region-sharded over the ranks; everything up to the engine calls on CPU ---
def test_strong_scaling_partition_two_ranks_under_torchrun():
    """generate (torch on the CPU) -> plan_genome -> slice per rank -> gloo gather: the code path of the real run up to the engine /
    RCCL calls.  The partition must be exact (every read owned once) and balanced."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29613", BENCH, "--gpus", "2", "--scaling", "strong", "--plan-only", "--strong-reads", "400000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _one_line(r.stdout)
    assert j["scaling"] == "strong" and j["world_seen"] == 2 and len(j["per_rank_reads"]) == 2
    assert sum(j["per_rank_reads"]) == j["reads"]
    assert j["imbalance_reads"] < 1.01
    assert all(l >= o for l, o in zip(j["per_rank_reads_loaded_with_halo"], j["per_rank_reads"]))      # halo reads are re-read, never dropped


def test_strong_scaling_plain_invocation_five_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "5", "--scaling", "strong", "--plan-only", "--strong-reads", "300000"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _one_line(r.stdout)
    assert j["world_seen"] == 5 and sum(j["per_rank_reads"]) == j["reads"] and j["imbalance_reads"] < 1.01
    assert sum(j["per_rank_pieces"]) >= 24          # every contig is somebody's, cut contigs twice


def test_plan_genome_covers_every_read_once():
    from metheor_amd import shard
    import random
    rnd = random.Random(7)
    for _ in range(200):
        contigs = [rnd.randrange(0, 5000) for _ in range(rnd.randrange(1, 30))]
        world = rnd.randrange(1, 65)
        plan = shard.plan_genome(contigs, world)
        assert len(plan) == world
        seen = [0] * len(contigs)
        prev = (-1, 0)
        for r in plan:
            for tid, a, b in r:
                assert 0 <= a < b <= contigs[tid]
                assert (tid, a) >= prev          # genome order, no overlap
                assert seen[tid] == a            # contiguous per contig
                seen[tid] = b
                prev = (tid, b)
        assert seen == contigs
        sizes = [sum(b - a for _, a, b in r) for r in plan]
        assert max(sizes) - min(sizes) <= 1


def test_order_line_puts_scalars_first_and_a_summary_last():
    """the bench line carries every secondary figure as a top-level scalar right behind the contract's keys and once
    more in a `summary` object at the END of the line (a driver log keeps the line's tail)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 2, "warmup": 1, "ms_per_step": 0.1, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": "w"},
           "roofline": {"frac": 0.43, "frac_l3_resident": 0.48}, "timed_region_s": 1.0,
           "all7": {"seven_measures_ms": 9.7, "per_pass_ms_one_sync_each": {"fdrp+qfdrp": 4.8, "mhl": 2.5}, "prepared_batches": {"seven_measures_ms": 8.8}},
           "fdrp_pairs": {"pass_ms": 3.6}, "e2e": {"M_reads_per_s_median": 40.0, "large": {"M_reads_per_s_median": 60.0}}, "cpu_baseline": {"value": 16.0}}
    line = bench.order_line(out)
    keys = list(line.keys())
    assert keys[:13] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]
    assert keys[-1] == "summary"
    first_nested = min(i for i, k in enumerate(keys) if isinstance(line[k], dict) and k != "config")
    for k in ("e2e_10M_median", "e2e_100M_median", "all7_ms", "all7_prepared_ms", "all7_fdrp_pass_ms", "fdrp_pairs_ms", "roofline_frac_two_batches"):
        assert k in line and keys.index(k) < first_nested and line["summary"][k] == line[k]
    assert line["all7_ms"] == 9.7 and line["e2e_100M_median"] == 60.0 and set(out) <= set(line)
