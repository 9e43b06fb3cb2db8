"""Byte-range sharding of a BAM without an index (mth_host_plan_shard): every shard gets a run of BGZF
blocks plus halo blocks and owns a (tid, pos) interval.  Checked against a plain-Python walk of the same file:
the owned intervals partition the genome, and every read a shard needs (start within halo_bp before its interval,
same contig, up to and including its end position) lies in the blocks it loads.  Also as 2 gloo ranks."""
import gzip
import os
import socket
import struct
import sys

import numpy as np
import pytest

from oracle import bamio
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INF = 2 ** 31 - 1


def make_bam(tmp_path, n_per=(3000, 1, 2500, 0, 1800), seed=3, unmapped=40):
    rng = np.random.default_rng(seed)
    refs = [("c%d" % k, 400_000) for k in range(len(n_per))]
    tid, pos = [], []
    for t, n in enumerate(n_per):
        p = np.sort(rng.integers(0, 399_000, n))
        if n > 200:
            p[50:90] = p[50]                   # many reads with one start: a run boundary may fall inside them
        tid += [t] * n
        pos += p.tolist()
    tid += [-1] * unmapped
    pos += [-1] * unmapped
    n = len(tid)
    cig = [[(100 << 4) | 0]] * n
    xms = [b"." * 100] * n
    rec = bamio.Records(refs, np.array(tid, np.int32), np.array(pos, np.int32), np.zeros(n, np.uint16),
                        np.full(n, 40, np.uint8), cig, xms)
    raw, bam = str(tmp_path / "raw.bam"), str(tmp_path / "aligned.bam")
    bamio.write_bam(raw, rec, realistic=True, seed=seed)
    util.reblock_aligned(raw, bam)
    return bam, rec


def record_blocks(bam, bz):
    """(tid, pos, block index) of every record, from a plain walk of the inflated stream"""
    raw = gzip.decompress(open(bam, "rb").read())
    ends = np.cumsum(bz["isize"].astype(np.int64))
    o = bz["header_bytes"]
    out = []
    while o < len(raw):
        bs, t, p = struct.unpack_from("<iii", raw, o)
        out.append((t if t >= 0 else INF, p, int(np.searchsorted(ends, o, side="right"))))
        o += 4 + bs
    return out


def check_plan(bam, world, halo):
    from metheor_amd import hostapi
    f = hostapi.BamFile(bam)
    bz = f.bgzf_blocks()
    recs = record_blocks(bam, bz)
    plans = [f.plan_shard(r, world, halo) for r in range(world)]
    f.close()
    assert plans[0]["beg"][0] == -1 and plans[-1]["end"][0] == INF
    for a, b in zip(plans, plans[1:]):
        assert a["end"] == b["beg"] or (a["end"][0] == INF and b["beg"][0] == INF)      # the intervals tile (tid, pos) space
    owned = 0
    for pl in plans:
        (tb, pb), (te, pe) = pl["beg"], pl["end"]
        if (tb, pb) == (te, pe):
            continue                              # more shards than runs: an empty interval needs nothing
        for (t, p, blk) in recs:
            own = (t, p) >= (tb, pb) and (t, p) < (te, pe)
            left = t == tb and pb - halo <= p < pb
            right = (t, p) == (te, pe)
            owned += own
            if own or left or right:
                assert pl["block_beg"] <= blk < pl["block_end"], (pl, t, p, blk)
        if pl["block_beg"] == 0 and pl["block_end"] > 0:
            assert pl["first_byte"] == bz["header_bytes"]
    assert owned == len(recs)                       # every record is owned by exactly one shard
    return plans


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8, 64])
def test_plan_covers_what_each_shard_needs(tmp_path, world):
    bam, _ = make_bam(tmp_path)
    plans = check_plan(bam, world, halo=500)
    if world in (2, 3):
        nb = max(p["block_end"] for p in plans)
        assert all(p["block_end"] - p["block_beg"] <= nb // world + 4 for p in plans)       # runs + a few halo blocks


def test_more_shards_than_blocks_and_empty_file(tmp_path, golden_dir):
    from metheor_amd import hostapi
    for k in (1, 5):
        f = hostapi.BamFile(os.path.join(golden_dir, "test%d.bam" % k))
        plans = [f.plan_shard(r, 4) for r in range(4)]
        f.close()
        assert plans[0]["beg"][0] == -1 and plans[-1]["end"][0] == INF
        assert sum(p["block_end"] > p["block_beg"] for p in plans) >= 1
    check_plan(os.path.join(golden_dir, "test1.bam"), 3, 100)


def test_unaligned_file_is_refused(tmp_path):
    from metheor_amd import hostapi
    bam, rec = make_bam(tmp_path)
    f = hostapi.BamFile(str(tmp_path / "raw.bam"))         # 60000-byte blocks cut through records
    with pytest.raises(hostapi.HostError):
        for r in range(8):
            f.plan_shard(r, 8)
    f.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, bam, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metheor_amd import hostapi
        f = hostapi.BamFile(bam)
        pl = f.plan_shard(rank, world, 500)
        f.close()
        mine = torch.tensor([pl["beg"][0], pl["beg"][1], pl["end"][0], pl["end"][1], pl["block_beg"], pl["block_end"]], dtype=torch.int64)
        allp = [torch.zeros(6, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allp, mine)
        if rank == 0:
            q.put([t.tolist() for t in allp])
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_plan_disjoint_ranges(tmp_path):
    """each rank plans only its own shard (no router, no exchange needed); gathered, the intervals tile the genome"""
    import torch.multiprocessing as mp
    bam, _ = make_bam(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, bam, q)) for r in range(2)]
    for p in ps:
        p.start()
    got = q.get(timeout=120)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (a, b) = got
    assert a[0] == -1 and b[2] == INF and a[2:4] == b[0:2]
    assert a[4] == 0 and a[5] < b[5] and b[4] > 0
