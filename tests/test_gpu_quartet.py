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


def test_wide_quartets_take_the_128_bit_key(eng):
    """consecutive CpGs of a read >= 2048 bp apart do not fit the packed 64-bit key (reference skips, big deletions, long
    reads): the reference has no such limit (readutil.rs:97-132), so these windows are aggregated under a 128-bit key
    (k_quartet_wide_insert) and come out as ordinary rows"""
    from metheor_amd import Batch, synth
    st = np.array([100], np.int32)
    pos = np.array([100, 200, 5000 | (1 << 31), 5100], np.uint32)
    b = Batch(0, 0, 100_000, st, st + 5100, np.array([42], np.uint8), np.array([0, 4], np.uint32), pos,
              np.array([0, 100, 4900, 5000], np.uint16))
    eng.reset()
    eng.quartet_accumulate(b)
    d = eng.quartet_fetch(0)
    assert d["pos"].tolist() == [[100, 200, 5000, 5100]]
    assert d["cnt"][0].tolist() == [0, 0, 1] + [0] * 13                      # pattern 0b0010: only the third CpG methylated
    # long sparse reads (most windows wide) on top of ordinary 150-bp reads, the same window seen by many reads
    rng = np.random.default_rng(404)
    sites = synth.make_sites(300_000, 0.0012, rng)
    long_c = synth.make_contig(0, 300_000, 900, 0.0012, rng, read_len=9000, sites=sites)
    n = np.diff(long_c["cpg_off"].astype(np.int64))
    gaps = np.diff(sites)
    assert (gaps >= 2048).sum() > 20 and n.max() >= 8
    short_c = synth.make_contig(0, 300_000, 40_000, 0.03, rng)
    # one coordinate-sorted batch holding both kinds
    def merged(a, b):
        order = np.argsort(np.concatenate([a["read_start"], b["read_start"]]), kind="stable")
        out = dict(a)
        cnt = np.concatenate([np.diff(a["cpg_off"].astype(np.int64)), np.diff(b["cpg_off"].astype(np.int64))])[order]
        off = np.zeros(len(order) + 1, np.int64); np.cumsum(cnt, out=off[1:])
        src_off = np.concatenate([a["cpg_off"][:-1].astype(np.int64), b["cpg_off"][:-1].astype(np.int64) + int(a["cpg_off"][-1])])[order]
        idx = np.repeat(src_off - off[:-1], cnt) + np.arange(off[-1])
        for k in ("read_start", "read_end", "read_mapq", "read_fwd"):
            out[k] = np.concatenate([a[k], b[k]])[order]
        out["cpg_off"] = off.astype(np.uint32)
        out["cpg_pos"] = np.concatenate([a["cpg_pos"], b["cpg_pos"]])[idx]
        out["cpg_rel"] = np.concatenate([a["cpg_rel"].astype(np.uint16), b["cpg_rel"].astype(np.uint16)])[idx]
        return out
    c = merged(long_c, short_c)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    for mq, md in ((10, 0), (0, 3)):
        d = run_device(eng, [c], mq, md)
        check(d, reads, mq, md)
    wide = (np.diff(d["pos"].astype(np.int64), axis=1) >= 2048).any(axis=1)
    assert wide.sum() > 20 and (~wide).sum() > 1000
    # and split into regions (a wide quartet is owned by the region of its first CpG)
    from metheor_amd import shard
    d2 = run_device(eng, [c], 0, 3, regions=[shard.plan_regions(c, 3)])
    check(d2, reads, 0, 3)


def test_table_overflow_retry(eng, monkeypatch):
    """the quartet table starts smaller than the number of quartet instances; an insert that runs out of probes makes the
    pass start over with a 4x larger table -- forced here by starting from 16 slots"""
    from metheor_amd import synth
    c = synth.make_contig(0, 200_000, 30_000, 0.05, np.random.default_rng(77))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    monkeypatch.setenv("MTH_QUARTET_SLOTS_MIN", "16")
    monkeypatch.setenv("MTH_QUARTET_FORCE_GLOBAL", "1")        # the global table only serves tiles the LDS table refused
    d = run_device(eng, [c], 10, 2)
    monkeypatch.delenv("MTH_QUARTET_SLOTS_MIN")
    monkeypatch.delenv("MTH_QUARTET_FORCE_GLOBAL")
    assert len(d["tid"]) > 2000
    check(d, reads, 10, 2)


def test_rows_come_out_sorted(eng):
    """tile path: rows are ordered by (tid, pos1..pos4) without any host sort (tiles in order, each sorted in LDS)"""
    from metheor_amd import synth
    rng = np.random.default_rng(5)
    cs = [synth.make_contig(0, 300_000, 60_000, 0.04, rng), synth.make_contig(1, 150_000, 30_000, 0.015, rng)]   # bitonic / rank sort
    d = run_device(eng, cs, 10, 0)
    order = np.lexsort((d["pos"][:, 3], d["pos"][:, 2], d["pos"][:, 1], d["pos"][:, 0], d["tid"]))
    assert len(order) > 5000 and (order == np.arange(len(order))).all()
    check(d, pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs)), 10, 0)


def test_global_path_for_every_tile(eng, monkeypatch):
    """every tile sent down the global-table path gives the same rows as the tile path"""
    from metheor_amd import shard, synth
    rng = np.random.default_rng(6)
    cs = [synth.make_contig(0, 300_000, 50_000, 0.05, rng), synth.make_contig(1, 100_000, 20_000, 0.2, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    monkeypatch.setenv("MTH_QUARTET_FORCE_GLOBAL", "1")
    d = run_device(eng, cs, 10, 3, regions=[shard.plan_regions(cs[0], 3), [(0, cs[1]["length"])]])
    monkeypatch.delenv("MTH_QUARTET_FORCE_GLOBAL")
    check(d, reads, 10, 3)
    assert len(d["tid"]) > 3000


def _reads_over(sites, starts, span, rng, p_meth=0.5):
    """reads [s, s+span) over a sorted site list: SoA dict in the synth contig layout"""
    lo = np.searchsorted(sites, starts)
    hi = np.searchsorted(sites, starts + span)
    n = (hi - lo).astype(np.uint32)
    off = np.zeros(len(starts) + 1, np.uint32)
    np.cumsum(n, out=off[1:])
    idx = np.repeat(lo - off[:-1].astype(np.int64), n) + np.arange(off[-1])
    pos = sites[idx].astype(np.uint32)
    meth = rng.random(len(pos)) < p_meth
    rel = (pos.astype(np.int64) - np.repeat(starts, n)).astype(np.uint8)
    return off, pos | (meth.astype(np.uint32) << 31), rel


def _contig(tid, length, starts, span, off, pos, rel, rng):
    n = len(starts)
    return {"tid": tid, "length": length, "read_start": starts.astype(np.int32),
            "read_end": (starts + span - 1).astype(np.int32), "read_mapq": rng.integers(0, 60, n).astype(np.uint8),
            "read_fwd": np.ones(n, np.uint8), "cpg_off": off, "cpg_pos": pos, "cpg_rel": rel}


def test_dense_tile_overflows_the_lds_table(eng):
    """a CpG every 2 bp: ~2000 distinct quartets start in one 4096-bp tile, more than the LDS table holds -> that tile
    (and only that one) takes the global path; the sparse tiles around it stay on the tile path"""
    from metheor_amd import synth
    rng = np.random.default_rng(8)
    dense = np.arange(8192, 8192 + 4096, 2)
    sparse = np.sort(rng.choice(np.r_[0:8192, 12288:60000], 1500, replace=False))
    sites = np.sort(np.r_[dense, sparse]).astype(np.int64)
    starts = np.sort(rng.integers(0, 60000 - 100, 20000)).astype(np.int64)
    off, pos, rel = _reads_over(sites, starts, 100, rng)
    c = _contig(0, 60000, starts, 100, off, pos, rel, rng)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    d = run_device(eng, [c], 10, 0)
    check(d, reads, 10, 0)
    p1 = d["pos"][:, 0]
    assert ((p1 >= 8192) & (p1 < 12288)).sum() > 1024
    # rows of the global path are merged into the batch's (pos1..pos4) order by the fetch: the output stays sorted
    assert (np.lexsort((d["pos"][:, 3], d["pos"][:, 2], d["pos"][:, 1], d["pos"][:, 0])) == np.arange(len(p1))).all()


def test_deep_tile_exceeds_16bit_bins(eng):
    """> 65535 candidate reads over one tile (deep amplicon): 16-bit bins would wrap, the tile takes the global path"""
    from metheor_amd import synth
    rng = np.random.default_rng(9)
    sites = np.array([5000, 5010, 5020, 5030, 5040, 9000, 9004, 9010, 9100, 9150], np.int64)
    starts = np.sort(np.r_[np.full(70_000, 4990), rng.integers(8950, 9000, 3000)]).astype(np.int64)
    off, pos, rel = _reads_over(sites, starts, 120, rng)
    c = _contig(0, 20000, starts, 120, off, pos, rel, rng)
    c["read_mapq"][:] = 40
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    d = run_device(eng, [c], 10, 0)
    check(d, reads, 10, 0)
    assert d["cnt"].sum(axis=1).max() == 70_000


def test_span_violation_is_refused(eng):
    """max_span smaller than a read's CpG extent: the tile path's candidate ranges would miss it -> MTH_ERR_SPAN"""
    from metheor_amd import Batch, MthError
    st = np.array([100], np.int32)
    pos = np.array([100, 120, 140, 400], np.uint32)
    b = Batch(0, 0, 100_000, st, st + 300, np.array([42], np.uint8), np.array([0, 4], np.uint32), pos,
              np.array([0, 20, 40, 300], np.uint16), max_span=150)
    eng.reset()
    eng.quartet_accumulate(b)
    with pytest.raises(MthError) as e:
        eng.quartet_fetch(0)
    assert e.value.status == -5
    eng.reset()


def test_output_redo_when_rows_do_not_fit(eng, monkeypatch):
    """the row buffers are sized from a guess; a batch that needs more is redone once with the exact size the kernel reports"""
    from metheor_amd import shard, synth
    rng = np.random.default_rng(10)
    cs = [synth.make_contig(0, 300_000, 50_000, 0.04, rng), synth.make_contig(1, 200_000, 30_000, 0.04, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    monkeypatch.setenv("MTH_QUARTET_ROWS_MIN", "100")
    d = run_device(eng, cs, 10, 0, regions=[shard.plan_regions(cs[0], 3), [(0, cs[1]["length"])]])
    monkeypatch.delenv("MTH_QUARTET_ROWS_MIN")
    check(d, reads, 10, 0)
    assert len(d["tid"]) > 10000


@pytest.mark.parametrize("shift", [13, 14, 15])
def test_every_tile_width(eng, monkeypatch, shift):
    """the tile width is chosen per batch from its call density; each of the three widths gives the same rows"""
    from metheor_amd import shard, synth
    rng = np.random.default_rng(40 + shift)
    cs = [synth.make_contig(0, 400_000, 60_000, 0.012, rng), synth.make_contig(1, 150_000, 20_000, 0.05, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    monkeypatch.setenv("MTH_QUARTET_TILE_SHIFT", str(shift))
    d = run_device(eng, cs, 10, 0, regions=[shard.plan_regions(cs[0], 2), [(0, cs[1]["length"])]])
    monkeypatch.delenv("MTH_QUARTET_TILE_SHIFT")
    check(d, reads, 10, 0)
    assert len(d["tid"]) > 7000


# ---- batches queued without a host sync (device-resident batches after the first of a context; quartet_batch / quartet_resolve) ----
def _queued_case(seed=31, n_contigs=5, length=60_000, reads=6_000):
    from metheor_amd import synth
    rng = np.random.default_rng(seed)
    cs = [synth.make_contig(t, length + length // 7 * t, reads + reads // 4 * t, 0.04, rng) for t in range(n_contigs)]
    recs = [util.contig_to_records(c, "ctg%d" % t) for t, c in enumerate(cs)]
    tabs = [(pyoracle.Reads.decode(r).me(min_depth=2), pyoracle.Reads.decode(r).pm(min_depth=2)) for r in recs]
    return cs, tabs


def _check_queued(d, tabs):
    n = 0
    for t, (om, op) in enumerate(tabs):
        m = d["tid"] == t
        n += int(m.sum())
        assert m.sum() == len(om)
        order = np.lexsort((d["pos"][m][:, 3], d["pos"][m][:, 2], d["pos"][m][:, 1], d["pos"][m][:, 0]))
        assert (d["pos"][m][order] == om.pos).all() and (d["cnt"][m][order] == om.cnt).all()
        assert (d["pm"][m][order].view(np.uint32) == op.val.view(np.uint32)).all()
        assert np.abs(d["me"][m][order].astype(np.float64) - om.val.astype(np.float64)).max() <= ME_TOL
    assert n == len(d["tid"])


@pytest.mark.parametrize("force", ["", "rows", "global"])
def test_queued_batches(eng, monkeypatch, capfd, force):
    """a job of device-resident batches: from the second batch on no call syncs; the fetch resolves them.  force = rows: the output is
    sized too small for every queued batch (they are replayed synchronously, exact size); global: every tile refuses its LDS table"""
    cs, tabs = _queued_case()
    monkeypatch.setenv("MTH_QUARTET_DEBUG", "1")
    eng.reset()
    bts = [util.device_batch(c, device="cuda:0") for c in cs]
    eng.quartet_accumulate(bts[0], min_qual=10)              # the first batch of a fresh context is synchronous whatever it is
    if force == "rows":
        eng.quartet_fetch(min_depth=2)                        # (capacity is only ever grown: start the check from a context that has little)
    if force == "global":
        monkeypatch.setenv("MTH_QUARTET_FORCE_GLOBAL", "1")
    for bt in bts[1:]:
        eng.quartet_accumulate(bt, min_qual=10)
    capfd.readouterr()
    d = eng.quartet_fetch(min_depth=2)
    err = capfd.readouterr().err
    import re
    q, r = map(int, re.search(r"\[quartet\] queued batches (\d+), replayed (\d+)", err).groups())
    assert q >= len(bts) - 1 and r == (len(bts) - 1 if force == "global" else 0), err        # (the first batch too, once the context has seen one)
    _check_queued(d, tabs)
    # the same job again after a reset, a count-only fetch first, then a batch more after the fetch (queue -> resolve -> queue)
    monkeypatch.delenv("MTH_QUARTET_FORCE_GLOBAL", raising=False)
    eng.reset()
    for bt in bts[:3]:
        eng.quartet_accumulate(bt, min_qual=10)
    d3 = eng.quartet_fetch(min_depth=2)
    _check_queued(d3, tabs[:3])
    for bt in bts[3:]:
        eng.quartet_accumulate(bt, min_qual=10)
    _check_queued(eng.quartet_fetch(min_depth=2), tabs)


def test_queued_batches_that_do_not_fit_are_replayed(monkeypatch, capfd):
    """a fresh context whose first (synchronous) batch is tiny: the output sizing learnt from it is far too small for the dense
    batches queued behind it -- they report 'did not fit' in their snapshots and the resolve replays them"""
    import metheor_amd
    from metheor_amd import synth
    rng = np.random.default_rng(5)
    tiny = synth.make_contig(0, 4_000, 40, 0.002, rng)
    cs, tabs = _queued_case(seed=77, n_contigs=3, length=1_500_000, reads=30_000)      # shallow and long: about one quartet per site
    monkeypatch.setenv("MTH_QUARTET_DEBUG", "1")
    e = metheor_amd.Engine(0)
    try:
        keep = [util.device_batch(tiny, device="cuda:0")] + [util.device_batch(dict(c, tid=c["tid"] + 1), device="cuda:0") for c in cs]
        for bt in keep:
            e.quartet_accumulate(bt, min_qual=10)
        capfd.readouterr()
        d = e.quartet_fetch(min_depth=2)
        err = capfd.readouterr().err
        assert "[quartet] queued batches 3, replayed" in err and "replayed 0" not in err, err
        m = d["tid"] >= 1
        _check_queued({k: (v[m] - 1 if k == "tid" else v[m]) for k, v in d.items()}, tabs)
    finally:
        e.close()


def test_queued_batch_beyond_its_estimate_is_replayed_not_truncated(monkeypatch, capfd):
    """a queued batch may produce more rows than its estimate and still fit the buffer (grown to 1.25 x the
    estimate); the next queued batch took the ESTIMATE as the rows in use and kept only that many when it grew the buffer -- the rows
    in between were lost without any flag.  Now a queued batch that exceeds its estimate is unfit and the resolve replays it.
    Batch B is given an estimate 10 % below its true row count (MTH_QUARTET_ROWS_MIN), batch C one that forces the buffer to grow."""
    import metheor_amd
    from metheor_amd import synth
    rng = np.random.default_rng(2024)
    shapes = [(100_000, 10_000), (600_000, 60_000), (900_000, 90_000)]
    cs = [synth.make_contig(t, ln, nr, 0.04, rng) for t, (ln, nr) in enumerate(shapes)]
    recs = [util.contig_to_records(c, "ctg%d" % t) for t, c in enumerate(cs)]
    tabs = [(pyoracle.Reads.decode(r).me(min_depth=2), pyoracle.Reads.decode(r).pm(min_depth=2)) for r in recs]
    rows_b = len(pyoracle.Reads.decode(recs[1]).me(min_depth=0))           # distinct quartets of batch B
    rows_c = len(pyoracle.Reads.decode(recs[2]).me(min_depth=0))
    monkeypatch.setenv("MTH_QUARTET_DEBUG", "1")
    e = metheor_amd.Engine(0)
    try:
        bts = [util.device_batch(c, device="cuda:0") for c in cs]
        e.quartet_accumulate(bts[0], min_qual=10)                           # synchronous: the context's first batch
        monkeypatch.setenv("MTH_QUARTET_ROWS_MIN", str(int(rows_b / 1.1)))  # B: estimate < rows <= 1.25 x estimate (fits the buffer)
        e.quartet_accumulate(bts[1], min_qual=10)
        monkeypatch.setenv("MTH_QUARTET_ROWS_MIN", str(2 * rows_c + 100_000))   # C: forces the buffer to grow
        e.quartet_accumulate(bts[2], min_qual=10)
        monkeypatch.delenv("MTH_QUARTET_ROWS_MIN")
        capfd.readouterr()
        d = e.quartet_fetch(min_depth=2)
        err = capfd.readouterr().err
        assert "[quartet] queued batches 2, replayed" in err and "replayed 0" not in err, err
        _check_queued(d, tabs)
    finally:
        e.close()
