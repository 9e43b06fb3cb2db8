"""N>1 path on CPU (world_size 2, gloo): what bench.py / a multi-GPU host does around the kernels.

Checked here: (1) metheor_amd.shard region planning + halo slicing gives every rank a batch whose
OWNED sites/reads partition the contig exactly, (2) the only exchange step -- all-reduce(sum) of the
4 LPMD int64 counters -- plus mth_lpmd_from_counts reproduces the single-process LPMD, (3) per-rank
PDR rows concatenated in rank order equal the single-process rows.  There is no GPU here, so each
rank's per-batch compute is done by the CPU oracle restricted to the rank's owned region (the
oracle is the stand-in for the device in THIS test only; the GPU tests check the device itself).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _contig():
    from metheor_amd import synth
    return synth.make_contig(0, 600_000, 90_000, 0.02, np.random.default_rng(42))


def _owned_rows(reads_tbl, beg, end):
    m = (reads_tbl.pos[:, 0] >= beg) & (reads_tbl.pos[:, 0] < end)
    return reads_tbl.pos[m, 0], reads_tbl.cnt[m], reads_tbl.val[m]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from metheor_amd import shard, synth
        from oracle import pyoracle
        c = _contig()
        regions = shard.plan_regions(c, world)
        beg, end = regions[rank]
        sub = shard.slice_region(c, beg, end)
        rd = pyoracle.Reads.from_soa(*synth.to_oracle_soa(sub))
        # PDR rows of the rank's OWNED sites (halo reads complete them)
        pos, cnt, val = _owned_rows(rd.pdr(min_depth=5, min_cpgs=2, min_qual=10), beg, end)
        # LPMD counters over the rank's OWNED reads only (start inside the region)
        own = (sub["read_start"] >= beg) & (sub["read_start"] < end)
        i0, i1 = int(np.argmax(own)) if own.any() else 0, int(len(own) - np.argmax(own[::-1])) if own.any() else 0
        assert own[i0:i1].all() and own.sum() == i1 - i0
        o0, o1 = int(sub["cpg_off"][i0]), int(sub["cpg_off"][i1])
        own_soa = (np.zeros(i1 - i0, np.int32), sub["read_start"][i0:i1], sub["read_end"][i0:i1],
                   sub["read_mapq"][i0:i1], sub["read_fwd"][i0:i1],
                   (sub["cpg_off"][i0:i1 + 1].astype(np.int64) - o0).astype(np.uint64),
                   sub["cpg_pos"][o0:o1], sub["cpg_rel"][o0:o1].astype(np.uint16))
        l = pyoracle.Reads.from_soa(*own_soa).lpmd(2, 16, 10)
        t = torch.tensor([l["n_concordant"], l["n_discordant"], l["n_read"], l["n_valid_read"]], dtype=torch.int64)
        dist.all_reduce(t)                                   # the one exchange step (RCCL on GPUs)
        rows = [None] * world
        dist.all_gather_object(rows, (pos, cnt, val))
        if rank == 0:
            q.put((t.tolist(), rows))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharding_and_lpmd_allreduce():
    import metheor_amd
    from metheor_amd import synth
    from oracle import pyoracle
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    counts, rows = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    c = _contig()
    ref = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    rl = ref.lpmd(2, 16, 10)
    assert counts == [rl["n_concordant"], rl["n_discordant"], rl["n_read"], rl["n_valid_read"]]
    L = metheor_amd.lib()
    assert np.float32(L.mth_lpmd_from_counts(counts[0], counts[1])) == np.float32(rl["lpmd"])
    rp = ref.pdr(min_depth=5, min_cpgs=2, min_qual=10)
    pos = np.concatenate([r[0] for r in rows]); cnt = np.concatenate([r[1] for r in rows]); val = np.concatenate([r[2] for r in rows])
    assert len(pos) == len(rp) > 1000
    assert (pos == rp.pos[:, 0]).all() and (cnt == rp.cnt).all()
    assert (val.view(np.uint32) == rp.val.view(np.uint32)).all()


def test_plan_regions_partition():
    from metheor_amd import shard
    c = _contig()
    for n in (1, 2, 3, 8):
        regs = shard.plan_regions(c, n)
        assert regs[0][0] == 0 and regs[-1][1] == c["length"]
        assert all(regs[k][1] == regs[k + 1][0] for k in range(n - 1))
        # every read start is owned by exactly one region; halo slices contain every read that can touch it
        owned = sum(int(((c["read_start"] >= b) & (c["read_start"] < e)).sum()) for b, e in regs)
        assert owned == len(c["read_start"])
        for b, e in regs:
            sub = shard.slice_region(c, b, e)
            touch = (c["read_end"] >= b) & (c["read_start"] - 1 < e)
            assert len(sub["read_start"]) >= int(touch.sum())
