"""The device path under SEVERAL PROCESSES (tests/test_multirank_gloo.py plans and reduces with the oracle as each
rank's compute).  One MI355X is all a test box has and RCCL refuses two ranks on one device, so the ranks share it
(`bench.py --share-devices`: a launcher mode whose line says "valid": false): each rank is its own process with its own device
context, runs the fused PDR + LPMD pass of ITS OWN contig on the GPU, the four LPMD counters are summed over the ranks (gloo on
host copies here; `mth_allreduce_lpmd_rank` when every rank has a device), rank 0 reports the maximum over the ranks.  Checked:
the job's n_read is the ranks' total (bench.py asserts it), world_seen, the per-rank times, ONE line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_processes_on_the_device(scaling):
    cmd = [sys.executable, BENCH, "--gpus", "2", "--share-devices", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-e2e", "--no-wgbs",
           "--no-traffic", "--soak-seconds", "0", "--preheat-seconds", "0"]
    cmd += ["--reads", "5000000"] if scaling == "weak" else ["--scaling", "strong", "--strong-reads", "4000000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_seen"] == 2 and j["scaling"] == scaling and j.get("valid") is False
    per = j["per_rank_ms_per_step"]
    assert len(per) == 2 and abs(j["ms_per_step"] - max(per)) < 1e-3 and j["value"] > 0
    if scaling == "strong":
        assert sum(j["per_rank_reads"]) == j["config"]["reads_total"]      # (and bench.py itself asserts that the ranks' n_read add up to it)
