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
