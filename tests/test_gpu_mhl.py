"""GPU parity of MHL (mth_mhl_accumulate / mth_mhl_fetch) against the CPU oracle, including the
reference's flush / re-open stream semantics (mhl.rs:162-173, 201-205).

Bar: sites and coverage bit-exact; MHL within 1e-6 absolute (mhl.rs:43-73 in f32; the oracle and
the device both sum over ascending l -- the reference's own order is HashMap-random)."""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 1e-6


@pytest.fixture(autouse=True, params=["tile", "walk", "tile-sub", "tile-hand-on", "tile-hand-on-big", "tile-rowchk", "tile-rowchk-sub", "tile-rowchk-32k", "tile-marks"])
def mhl_form(request, monkeypatch):
    """every case runs through the one-pass tile form (mth_mhl_tile.hip, the default), through round 2's discovery + per-site walk
    (MTH_MHL_WALK=1), through the tile form started with 256-position sub-ranges (the path a tile with more sites than slots
    takes) and through the tile form with every site handed on to the exact walks (the path of sites with several segments): the
    wave-per-site walk (k_mhl_walk_wave), or -- "big" -- the lane-per-site walk that takes what the wave walk leaves; and with the
    flush rule looked at per finished row (k_mhl_rowcheck, round 6: the host's choice for batches of <= 2 CpGs a read) or per read
    (the tile kernel's position bitmap) whatever the batch's density"""
    for k in ("MTH_MHL_WALK", "MTH_MHL_FORCE_SUB", "MTH_MHL_FORCE_HAND_ON", "MTH_MHL_NO_WAVE_WALK", "MTH_MHL_ROWCHK", "MTH_MHL_TILE_SHIFT"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "walk":
        monkeypatch.setenv("MTH_MHL_WALK", "1")
    elif request.param == "tile-sub":
        monkeypatch.setenv("MTH_MHL_FORCE_SUB", "1")
    elif request.param == "tile-hand-on":
        monkeypatch.setenv("MTH_MHL_FORCE_HAND_ON", "1")
    elif request.param == "tile-rowchk":
        monkeypatch.setenv("MTH_MHL_ROWCHK", "1")
    elif request.param == "tile-rowchk-sub":
        monkeypatch.setenv("MTH_MHL_ROWCHK", "1")
        monkeypatch.setenv("MTH_MHL_FORCE_SUB", "1")
    elif request.param == "tile-rowchk-32k":        # the 32 768-position tiles the host takes on large sparse-call regions
        monkeypatch.setenv("MTH_MHL_ROWCHK", "1")
        monkeypatch.setenv("MTH_MHL_TILE_SHIFT", "15")
    elif request.param == "tile-marks":
        monkeypatch.setenv("MTH_MHL_ROWCHK", "0")
    elif request.param == "tile-hand-on-big":
        monkeypatch.setenv("MTH_MHL_FORCE_HAND_ON", "1")
        monkeypatch.setenv("MTH_MHL_NO_WAVE_WALK", "1")
    return request.param


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


def run_device(eng, contigs, kw, device=None, regions=None):
    from metheor_amd import shard
    eng.reset()
    keep = []
    for ci, c in enumerate(contigs):
        regs = regions[ci] if regions else [(0, c["length"])]
        for (b, e) in regs:
            sub = shard.slice_region(c, b, e) if regions else c
            bt = util.device_batch(sub, region=(b, e), device=device)
            keep.append(bt)
            eng.mhl_accumulate(bt, **kw)
    return eng.mhl_fetch()


def check(dev, reads, kw):
    o = reads.mhl(**kw)
    assert len(dev["pos"]) == len(o), (len(dev["pos"]), len(o))
    assert (dev["tid"] == o.tid).all() and (dev["pos"] == o.pos[:, 0]).all()
    assert (dev["cov"] == o.cnt[:, 0]).all()
    diff = np.abs(dev["mhl"].astype(np.float64) - o.val.astype(np.float64))
    assert len(diff) == 0 or diff.max() <= TOL, diff.max()
    return int((dev["mhl"].view(np.uint32) != o.val.view(np.uint32)).sum()), len(o)


def soa_contig(reads_rows, length=100000):
    """rows: (start, mapq, fwd, [(rel, pos, meth), ...]) -> contig dict + oracle Reads"""
    start = np.array([r[0] for r in reads_rows], np.int32)
    ends = np.array([r[0] + 99 for r in reads_rows], np.int32)
    mapq = np.array([r[1] for r in reads_rows], np.uint8)
    fwd = np.array([r[2] for r in reads_rows], np.uint8)
    off = np.zeros(len(reads_rows) + 1, np.uint32)
    pos, rel = [], []
    for i, r in enumerate(reads_rows):
        for (rl, p, m) in r[3]:
            pos.append(p | (int(m) << 31)); rel.append(rl)
        off[i + 1] = len(pos)
    c = dict(tid=0, length=length, read_start=start, read_end=ends, read_mapq=mapq, read_fwd=fwd, cpg_off=off,
             cpg_pos=np.array(pos, np.uint32), cpg_rel=np.array(rel, np.uint8))
    from metheor_amd import synth
    return c, pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))


# ---- the reference's fixtures and known answers (mhl.rs:243-295) ------------------------------------
@pytest.mark.parametrize("k,nsite,want", [(1, 4, 0.1625), (2, 4, 0.5), (3, 4, 0.5), (4, 8, 0.1625), (5, 0, None)])
def test_reference_fixtures(eng, golden_dir, k, nsite, want):
    rec = bamio.read_bam(os.path.join(golden_dir, "test%d.bam" % k))
    reads = pyoracle.Reads.decode(rec)
    c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
    kw = dict(min_depth=0, min_cpgs=0, min_qual=10)
    d = run_device(eng, [c], kw)
    assert len(d["pos"]) == nsite
    if nsite:
        assert (d["mhl"] == f32(want)).all()
    check(d, reads, kw)
    if k == 1:   # `metheor mhl` defaults on test1 -> 4 sites, 0.1625
        d = run_device(eng, [c], dict(min_depth=10, min_cpgs=4, min_qual=10))
        assert d["pos"].tolist() == [0, 2, 4, 6] and (d["mhl"] == f32(0.1625)).all() and (d["cov"] == 16).all()


def test_reopen_semantics_hand_built(eng):
    """A forward read that starts ON the G of CpG c (so its first CpG lies after c) flushes c; a
    reverse read with the same start then reports c at start-1 and RE-OPENS it (mhl.rs:162-173).
    The last segment with coverage >= min_depth wins; a later, too-shallow segment does not erase it."""
    site = 1000
    A = lambda s, calls: (s, 40, 1, calls)
    rows = []
    for _ in range(3):   # three reads covering c=1000 and 1010, fully methylated
        rows.append(A(990, [(10, site, 1), (20, 1010, 1)]))
    rows.append(A(1001, [(9, 1010, 0), (19, 1020, 0)]))                      # flusher: first CpG 1010 > 1000
    rows.append((1001, 40, 0, [(0, site, 0), (10, 1010, 0)]))                # reverse: re-opens 1000, unmethylated
    rows.append((1001, 40, 0, [(0, site, 0), (10, 1010, 0)]))
    c, reads = soa_contig(rows)
    for md in (0, 2, 3):
        kw = dict(min_depth=md, min_cpgs=0, min_qual=10)
        d = run_device(eng, [c], kw)
        check(d, reads, kw)
        row = {int(p): (float(v), int(cv)) for p, v, cv in zip(d["pos"], d["mhl"], d["cov"])}
        if md <= 2:
            assert row[site] == (0.0, 2)          # second segment (2 unmethylated reads) overwrote the first
        else:
            assert row[site][1] == 3 and row[site][0] > 0.9   # second segment too shallow: first one stands
    # a read failing mapq still flushes (flush happens BEFORE the filters, mhl.rs:162 vs 176)
    rows2 = [A(990, [(10, site, 1), (20, 1010, 1)]), (1001, 0, 1, [(9, 1010, 0), (19, 1020, 0)]),
             (1001, 40, 0, [(0, site, 0), (10, 1010, 0)])]
    c2, reads2 = soa_contig(rows2)
    kw = dict(min_depth=0, min_cpgs=0, min_qual=10)
    d = run_device(eng, [c2], kw)
    check(d, reads2, kw)
    assert {int(p): int(cv) for p, cv in zip(d["pos"], d["cov"])}[site] == 1


def test_real_rrbs_reads(eng, golden_dir):
    rec = bamio.read_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    reads = pyoracle.Reads.decode(rec)
    c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
    for kw in (dict(min_depth=0, min_cpgs=0, min_qual=10), dict(min_depth=10, min_cpgs=4, min_qual=10),
               dict(min_depth=3, min_cpgs=2, min_qual=43), dict(min_depth=1, min_cpgs=1, min_qual=0)):
        ne, n = check(run_device(eng, [c], kw), reads, kw)
        print("RRBS", kw, "rows", n, "not bit-identical", ne)


@pytest.mark.parametrize("device_mem", [False, True])
def test_synthetic_vs_oracle(eng, device_mem):
    """50/50 strands: error-free data re-opens ~0.3 % of sites under MHL's strict flush"""
    from metheor_amd import synth
    c = synth.make_contig(1, 1_500_000, 250_000, 0.02, np.random.default_rng(41))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    dev = "cuda:0" if device_mem else None
    for kw in (dict(min_depth=10, min_cpgs=4, min_qual=10), dict(min_depth=0, min_cpgs=0, min_qual=0), dict(min_depth=5, min_cpgs=1, min_qual=10)):
        d = run_device(eng, [c], kw, device=dev)
        ne, n = check(d, reads, kw)
        print("synthetic", kw, "rows", n, "not bit-identical", ne)
        assert n > 1000
    # the data really contains re-opened sites: plain per-site pooling (no flush) differs somewhere
    kw = dict(min_depth=0, min_cpgs=0, min_qual=0)
    o = reads.mhl(**kw)
    off = c["cpg_off"].astype(np.int64)
    pos = (c["cpg_pos"] & 0x7fffffff).astype(np.int64)
    pooled = np.bincount(pos, minlength=c["length"])[o.pos[:, 0]]
    assert (pooled != o.cnt[:, 0]).sum() > 0


def test_multi_contig_region_split_and_dense(eng):
    from metheor_amd import shard, synth
    rng = np.random.default_rng(43)
    cs = [synth.make_contig(0, 200_000, 30_000, 0.03, rng), synth.make_contig(1, 400_000, 80_000, 0.03, rng),
          synth.make_contig(2, 60_000, 12_000, 0.2, rng)]           # ~25 CpGs/read: the big (scratch) variant
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    kw = dict(min_depth=5, min_cpgs=2, min_qual=10)
    regions = [[(0, cs[0]["length"])], shard.plan_regions(cs[1], 4), shard.plan_regions(cs[2], 2)]
    d = run_device(eng, cs, kw, regions=regions)
    check(d, reads, kw)
    d2 = run_device(eng, cs, kw)
    check(d2, reads, kw)
    assert (d["pos"] == d2["pos"]).all() and (d["mhl"].view(np.uint32) == d2["mhl"].view(np.uint32)).all()
    assert (d["tid"] == 2).sum() > 500


def test_reset_empty_and_capacity(eng):
    from metheor_amd import Batch, MthError, synth
    eng.reset()
    z4 = np.zeros(0, np.int32)
    b = Batch(0, 0, 1000, z4, z4, np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    eng.mhl_accumulate(b)
    assert len(eng.mhl_fetch()["pos"]) == 0
    # reads with > 16384 CpGs are beyond the histogram capacity: loud MTH_ERR_CAPACITY, never a wrong value
    c = synth.make_contig(0, 400_000, 6, 0.45, np.random.default_rng(5), read_len=60_000)
    assert np.diff(c["cpg_off"].astype(np.int64)).max() > 16384
    eng.reset()
    eng.mhl_accumulate(util.device_batch(c), min_depth=0, min_cpgs=0, min_qual=0)
    with pytest.raises(MthError) as e:
        eng.mhl_fetch()
    assert e.value.status == -8
    eng.reset()


def test_long_reads_beyond_512_cpgs(eng):
    """mhl.rs:185-192 has no limit on a read's CpG count: reads with up to 16384 CpGs take the walk whose histograms live in
    HBM scratch (k_mhl_walk_huge); rows equal to the oracle's, mixed with ordinary reads in the same batch"""
    from metheor_amd import synth
    rng = np.random.default_rng(77)
    long_c = synth.make_contig(0, 120_000, 70, 0.3, rng, read_len=6000)          # ~1800 CpGs per read
    n = np.diff(long_c["cpg_off"].astype(np.int64))
    assert n.max() > 1000 and n.max() < 16384
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(long_c))
    for kw in (dict(min_depth=0, min_cpgs=0, min_qual=0), dict(min_depth=3, min_cpgs=4, min_qual=10)):
        d = run_device(eng, [long_c], kw)
        nd, nrows = check(d, reads, kw)
        assert nrows > 1000


def test_deep_stretch_queue_overflow_and_heavy(eng):
    """thousands-fold depth on a few hundred bp (amplicon data): the tile form's contributor queue (1024 reads per stretch)
    overflows, the stretch is halved down to 256 positions and then only counted and handed on to the exact walk; with more than
    8191 candidate reads the 16-bit bins are not used at all.  Same rows as the oracle either way."""
    from metheor_amd import synth
    rng = np.random.default_rng(47)
    for n_reads in (3000, 12000):
        # all the reads start within 300 bp in the middle of the contig
        st = np.sort(rng.integers(10_000, 10_300, n_reads)).astype(np.int32)
        deep = synth.make_contig(0, 20_000, n_reads, 0.04, rng, starts=st)
        reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(deep))
        kw = dict(min_depth=10, min_cpgs=4, min_qual=10)
        d = run_device(eng, [deep], kw)
        ne, n = check(d, reads, kw)
        assert n > 5
