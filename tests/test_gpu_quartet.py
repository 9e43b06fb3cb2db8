"""GPU parity of ME / PM (mth_quartet_accumulate / mth_quartet_fetch) against the CPU oracle.

Bar: 16-bin histograms and quartet keys bit-exact; PM bit-exact (pm.rs:42-51 on the same integers,
no FMA contraction on the device); ME within 1e-6 absolute (me.rs:42-55 uses log2f)."""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu
f32 = np.float32
ME_TOL = 1e-6


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


def run_device(eng, contigs, min_qual, min_depth, device=None, regions=None):
    from metheor_amd import shard
    eng.reset()
    keep = []
    for ci, c in enumerate(contigs):
        regs = regions[ci] if regions else [(0, c["length"])]
        for (b, e) in regs:
            sub = shard.slice_region(c, b, e) if regions else c
            bt = util.device_batch(sub, region=(b, e), device=device)
            keep.append(bt)
            eng.quartet_accumulate(bt, min_qual=min_qual)
    return eng.quartet_fetch(min_depth=min_depth)


def check(dev, reads, min_qual, min_depth):
    om, op = reads.me(min_depth=min_depth, min_qual=min_qual), reads.pm(min_depth=min_depth, min_qual=min_qual)
    assert len(dev["tid"]) == len(om) == len(op)
    order = np.lexsort((dev["pos"][:, 3], dev["pos"][:, 2], dev["pos"][:, 1], dev["pos"][:, 0], dev["tid"]))
    assert (dev["tid"][order] == om.tid).all()
    assert (dev["pos"][order] == om.pos).all()
    assert (dev["cnt"][order] == om.cnt).all()                                   # histograms bit-exact
    assert (dev["pm"][order].view(np.uint32) == op.val.view(np.uint32)).all()    # PM bit-exact
    diff = np.abs(dev["me"][order].astype(np.float64) - om.val.astype(np.float64))
    assert len(diff) == 0 or diff.max() <= ME_TOL, diff.max()
    return float(diff.max()) if len(diff) else 0.0


# ---- the reference's fixtures and known answers (me.rs:138-207, pm.rs:134-202) --------------------
@pytest.mark.parametrize("k,nq,me,pm", [(1, 1, 1.0, 0.9375), (2, 1, 0.25, 0.5), (3, 1, 0.25, 0.5), (4, 2, 1.0, 0.9375), (5, 0, None, None)])
def test_reference_fixtures(eng, golden_dir, k, nq, me, pm):
    rec = bamio.read_bam(os.path.join(golden_dir, "test%d.bam" % k))
    reads = pyoracle.Reads.decode(rec)
    c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
    d = run_device(eng, [c], min_qual=10, min_depth=0)
    assert len(d["tid"]) == nq
    if nq:
        assert (d["me"] == f32(me)).all() and (d["pm"] == f32(pm)).all()
    if k == 1:
        assert d["pos"].tolist() == [[0, 2, 4, 6]] and (d["cnt"] == 1).all()     # depth 16, every pattern once (me.rs:149)
    check(d, reads, 10, 0)
    # the write-time depth filter (me.rs:82): default -d 10 keeps test1's quartet (depth 16), -d 17 drops it
    assert len(run_device(eng, [c], 10, 10)["tid"]) == (nq if k != 3 else 0)
    assert len(run_device(eng, [c], 10, 17)["tid"]) == 0


def test_real_rrbs_reads(eng, golden_dir):
    rec = bamio.read_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    reads = pyoracle.Reads.decode(rec)
    c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
    worst = 0.0
    for mq, md in ((10, 0), (10, 10), (43, 0), (0, 3)):
        worst = max(worst, check(run_device(eng, [c], mq, md), reads, mq, md))
    assert len(run_device(eng, [c], 10, 0)["tid"]) > 20
    print("max |ME_dev - ME_oracle| on RRBS:", worst)


@pytest.mark.parametrize("device_mem", [False, True])
def test_synthetic_vs_oracle(eng, device_mem):
    from metheor_amd import synth
    c = synth.make_contig(2, 1_000_000, 200_000, 0.05, np.random.default_rng(31))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    dev = "cuda:0" if device_mem else None
    worst = 0.0
    for mq, md in ((10, 10), (0, 0)):
        d = run_device(eng, [c], mq, md, device=dev)
        worst = max(worst, check(d, reads, mq, md))
        assert len(d["tid"]) > 5000
    print("max |ME_dev - ME_oracle| synthetic:", worst)


def test_multi_contig_region_split_and_dense(eng):
    """quartets are owned by the region holding pos1: split batches give the same rows; dense CpGs"""
    from metheor_amd import shard, synth
    rng = np.random.default_rng(33)
    cs = [synth.make_contig(0, 200_000, 30_000, 0.04, rng), synth.make_contig(1, 500_000, 90_000, 0.04, rng),
          synth.make_contig(2, 100_000, 20_000, 0.2, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    regions = [[(0, cs[0]["length"])], shard.plan_regions(cs[1], 4), shard.plan_regions(cs[2], 3)]
    d = run_device(eng, cs, 10, 5, regions=regions)
    check(d, reads, 10, 5)
    d2 = run_device(eng, cs, 10, 5)
    check(d2, reads, 10, 5)
    assert len(d["tid"]) == len(d2["tid"]) > 10000


def test_reset_and_empty(eng):
    from metheor_amd import Batch
    eng.reset()
    z4 = np.zeros(0, np.int32)
    b = Batch(0, 0, 1000, z4, z4, np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    eng.quartet_accumulate(b)
    assert len(eng.quartet_fetch(0)["tid"]) == 0
    eng.reset()
    assert len(eng.quartet_fetch(0)["tid"]) == 0


def test_wide_quartet_is_refused(eng):
    """consecutive CpGs of a read >= 2048 bp apart do not fit the packed key: loud MTH_ERR_CAPACITY"""
    from metheor_amd import Batch, MthError
    st = np.array([100], np.int32)
    pos = np.array([100, 200, 5000, 5100], np.uint32)
    b = Batch(0, 0, 100_000, st, st + 5100, np.array([42], np.uint8), np.array([0, 4], np.uint32), pos,
              np.array([0, 100, 4900, 5000], np.uint16))
    eng.reset()
    eng.quartet_accumulate(b)
    with pytest.raises(MthError) as e:
        eng.quartet_fetch(0)
    assert e.value.status == -8
    eng.reset()


def test_table_overflow_retry(eng, monkeypatch):
    """the quartet table starts smaller than the number of quartet instances; an insert that runs out of probes makes the
    pass start over with a 4x larger table -- forced here by starting from 16 slots"""
    from metheor_amd import synth
    c = synth.make_contig(0, 200_000, 30_000, 0.05, np.random.default_rng(77))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    monkeypatch.setenv("MTH_QUARTET_SLOTS_MIN", "16")
    d = run_device(eng, [c], 10, 2)
    monkeypatch.delenv("MTH_QUARTET_SLOTS_MIN")
    assert len(d["tid"]) > 2000
    check(d, reads, 10, 2)
