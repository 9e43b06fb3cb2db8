"""The exchange step between DISTINCT devices (lpmd.rs:11-12, 51-55: the four genome-wide LPMD counters are the only
thing the region shards share).  Every test here needs at least two visible GPUs and skips on the one-GPU boxes the rounds have been
tested on so far; on the first multi-GPU lease they are the first execution of RCCL between two devices WITH an assertion behind it:

* `mth_allreduce_lpmd` over two contexts on two devices in one process (ncclCommInitAll) against the oracle's counters;
* `mth_rccl_init_rank` + `mth_allreduce_lpmd_rank`, one process per GPU under `torch.distributed.run --nproc-per-node 2` -- once,
  and 13 times in a row (more reduces than the ring of slots holds: the reuse of a slot behind its done-event);
* `metheor lpmd --gpus 2` and `metheor pdr --gpus 2` with the shards on distinct devices, against the oracle's text;
* `bench.py --gpus 2`, weak and strong, `world_seen == 2` and no `"valid": false` in the line (that key marks the shared-device
  launcher test of tests/test_gpu_multirank.py).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metheor_amd", "metheor")
BENCH = os.path.join(ROOT, "bench.py")


def _ndev():
    try:
        import metheor_amd
        return metheor_amd.device_count()
    except Exception:
        return 0


need2 = pytest.mark.skipif(_ndev() < 2, reason="needs two visible GPUs (RCCL refuses two ranks on one device)")


def _two_contigs(seed=5):
    from metheor_amd import synth
    rng = np.random.default_rng(seed)
    return [synth.make_contig(0, 400_000, 60_000, 0.03, rng), synth.make_contig(1, 250_000, 35_000, 0.03, rng)]


@need2
def test_allreduce_between_two_devices_one_process():
    """one Engine per device, each accumulates ITS contig; one ncclAllReduce(int64 x 4, sum); both report the oracle's totals"""
    import metheor_amd
    from metheor_amd import PdrLpmdParams, synth
    cs = _two_contigs()
    want = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs)).lpmd()
    engs = [metheor_amd.Engine(d) for d in (0, 1)]
    try:
        keep = []
        for d, (e, c) in enumerate(zip(engs, cs)):
            bt = util.device_batch(c, device="cuda:%d" % d)
            keep.append(bt)
            e.reset()
            e.pdr_lpmd_accumulate(bt, PdrLpmdParams())
        own = [e.lpmd_global() for e in engs]
        assert own[0]["n_read"] == len(cs[0]["read_start"]) and own[1]["n_read"] == len(cs[1]["read_start"])
        metheor_amd.allreduce_lpmd(engs)
        for e in engs:
            g = e.lpmd_global()
            for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"):
                assert g[k] == want[k], (k, g[k], want[k])
            assert np.float32(g["lpmd"]).view(np.uint32) == np.float32(want["lpmd"]).view(np.uint32)
    finally:
        for e in engs:
            e.close()


_RANK_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import metheor_amd
from metheor_amd import PdrLpmdParams, synth
from tests import util
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(local)
dist.init_process_group("gloo")          # only to ship the 128-byte RCCL id; the reduce itself is mth_allreduce_lpmd_rank
rng = np.random.default_rng(5)
cs = [synth.make_contig(0, 400_000, 60_000, 0.03, rng), synth.make_contig(1, 250_000, 35_000, 0.03, rng)]
uid = [metheor_amd.Engine.rccl_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
e = metheor_amd.Engine(local)
e.rccl_init_rank(uid[0], rank, world)
bt = util.device_batch(cs[rank], device="cuda:%%d" %% local)
e.reset(); e.pdr_lpmd_accumulate(bt, PdrLpmdParams())
e.allreduce_lpmd_rank()
g = e.lpmd_global()
print("RANK_RESULT " + json.dumps({"rank": rank, "device": local, **{k: int(g[k]) for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read")}}), flush=True)
e.close()
dist.destroy_process_group()
'''


_RING_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import metheor_amd
from metheor_amd import PdrLpmdParams, synth
from tests import util
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", rank))
torch.cuda.set_device(local)
dist.init_process_group("gloo")
rng = np.random.default_rng(5)
cs = [synth.make_contig(0, 400_000, 60_000, 0.03, rng), synth.make_contig(1, 250_000, 35_000, 0.03, rng)]
uid = [metheor_amd.Engine.rccl_unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
e = metheor_amd.Engine(local)
e.rccl_init_rank(uid[0], rank, world)
bt = util.device_batch(cs[rank], device="cuda:%%d" %% local)
e.reset()
K = ("n_concordant", "n_discordant", "n_read", "n_valid_read")
# (a) more reduces than the ring has slots (RED_RING = 4), back to back, nothing read in between: every slot is reused behind its
#     done-event while the next batches' kernels are queued; the counters keep running (no reset), so the last reduce holds 7 x the totals
for it in range(7):
    e.pdr_lpmd_accumulate(bt, PdrLpmdParams())
    e.allreduce_lpmd_rank()
g = e.lpmd_global()
print("RING_RESULT " + json.dumps({"rank": rank, "phase": "a", "mult": 7, **{k: int(g[k]) for k in K}}), flush=True)
# (b) a read-back after every reduce (each drains the side stream): 8 x, 9 x, ... 13 x
for it in range(6):
    e.pdr_lpmd_accumulate(bt, PdrLpmdParams())
    e.allreduce_lpmd_rank()
    g = e.lpmd_global()
    print("RING_RESULT " + json.dumps({"rank": rank, "phase": "b", "mult": 8 + it, **{k: int(g[k]) for k in K}}), flush=True)
e.close()
dist.destroy_process_group()
'''


@need2
def test_allreduce_rank_form_ring_slot_reuse(tmp_path):
    """mth_allreduce_lpmd_rank reduces out of place in a ring of RED_RING = 4 slots on a side stream (mth_rccl.hip: a slot is
    reused behind its done-event) -- the one intricate part of the exchange step.  More reduces than slots, back to back and with a
    read-back after each; the k-th reduce must hold k x the two ranks' totals (the counters are not reset in between)."""
    from metheor_amd import synth
    script = tmp_path / "ring.py"
    script.write_text(_RING_SCRIPT % dict(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29633", str(script)], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    got = [json.loads(l.split(" ", 1)[1]) for l in r.stdout.splitlines() if l.startswith("RING_RESULT ")]
    assert len(got) == 2 * 7
    want = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(_two_contigs())).lpmd()
    for g in got:
        for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"):
            assert g[k] == g["mult"] * want[k], (g, k, want[k])


@need2
def test_allreduce_rank_form_under_torchrun(tmp_path):
    """one process per GPU (the launch shape bench.py --gpus N and a Rust host with one process per device would use)"""
    from metheor_amd import synth
    script = tmp_path / "rank.py"
    script.write_text(_RANK_SCRIPT % dict(root=ROOT))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", str(script)], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    got = [json.loads(l.split(" ", 1)[1]) for l in r.stdout.splitlines() if l.startswith("RANK_RESULT ")]
    assert sorted(g["rank"] for g in got) == [0, 1] and sorted(g["device"] for g in got) == [0, 1]
    want = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(_two_contigs())).lpmd()
    for g in got:
        for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"):
            assert g[k] == want[k], (g, k, want[k])


@need2
@pytest.mark.parametrize("sub,extra", [("lpmd", []), ("pdr", ["-d", "3", "-p", "1"])])
def test_cli_two_shards_on_two_devices(tmp_path, sub, extra):
    """`metheor <sub> --gpus 2` with the shards' contexts on devices 0 and 1 (shard r runs on device r mod the visible devices):
    byte-identical to the single run and equal to the oracle's text"""
    rec_parts = []
    refs = [("sA", 400_000), ("sB", 250_000)]
    tid, pos, flag, mapq, cig, xms = [], [], [], [], [], []
    for t, c in enumerate(_two_contigs()):
        r = util.contig_to_records(c, refs[t][0])
        tid += [t] * len(r); pos += r.pos.tolist(); flag += r.flag.tolist(); mapq += r.mapq.tolist(); cig += r.cigars; xms += r.xms
    rec = bamio.Records(refs, tid, pos, flag, mapq, cig, xms)
    bam = str(tmp_path / "two.bam")
    bamio.write_bam(bam, rec)
    o1, o2 = tmp_path / "one.tsv", tmp_path / "two.tsv"
    env = dict(os.environ, METHEOR_SHARD_HALO="4000")
    for out, more in ((o1, []), (o2, ["--gpus", "2"])):
        r = subprocess.run([EXE, sub, "-i", bam, "-o", str(out)] + extra + more, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
    assert o1.read_bytes() == o2.read_bytes() and len(o1.read_bytes()) > 20
    reads = pyoracle.Reads.decode(rec)
    want, _ = util.oracle_text(reads, [n for n, _ in refs], sub, input_name=bam, seed=0, **util.oracle_kwargs(sub, extra))
    util.assert_tsv_equals_oracle(sub, o2.read_text(), want)


@need2
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_gpus(scaling):
    """the driver's own launch shape at N = 2: one rank per device, RCCL for the counters, ONE line, `world_seen` 2, nothing marking it invalid"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29633",
           BENCH, "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-e2e", "--no-wgbs", "--no-traffic", "--soak-seconds", "0",
           "--preheat-seconds", "0"]
    cmd += ["--reads", "2000000"] if scaling == "weak" else ["--scaling", "strong", "--strong-reads", "4000000"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["world_seen"] == 2 and j["scaling"] == scaling and "valid" not in j
    assert len(j["per_rank_ms_per_step"]) == 2 and j["value"] > 0


def test_skip_reason_is_the_device_count():
    """on a one-GPU box this file must skip, not fail: the marker above is the only gate"""
    assert _ndev() >= 1
