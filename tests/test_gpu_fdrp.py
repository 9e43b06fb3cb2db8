"""GPU parity of FDRP / qFDRP (mth_fdrp_accumulate / mth_fdrp_fetch) against the CPU oracle.

Bar: sites and stored-read counts bit-exact; FDRP/qFDRP within 1e-6 absolute.  Where depth exceeds
max_depth the reference samples with an OS-seeded RNG (fdrp.rs:90) -- nothing can be bit-equal to it;
device and oracle share a counter-based draw, so they are compared with each other (the branch is
"parity unpinned" against the reference)."""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu
f32 = np.float32
TOL = 1e-6


@pytest.fixture(autouse=True, params=["auto", "wtile", "wtile_sub", "wtile_heavy", "walk16", "walk32", "tile", "tile_wide_rows", "walk"])
def walk_choice(request, monkeypatch):
    """every case eight times: with the host's own choice of kernel form; with the one-pass tile form forced (k_fdrp_wtile,
    round 6: whatever it cannot hold is handed back to the general walk) -- as it is, starting from 192-position stretches, and
    with every stretch on its count-only path (every site handed back); with k_fdrp_walk4 forced (16 or 32 lanes per site, the
    rest handed back to the general walk -- on dense data mostly the hand-back path); with the read x read form forced
    (k_fdrp_tile + k_fdrp_chain, round 4: its stored reads' rows in the four-word form where every reader spans <= 16 window sites
    -- round 6 -- and in the eight-word form forced); and with the wave-per-site walk alone"""
    for k in ("METHEOR_FDRP_WALK4", "METHEOR_FDRP_TILE", "METHEOR_FDRP_WTILE", "METHEOR_FDRP_WTILE_SUB", "METHEOR_FDRP_WTILE_HEAVY", "METHEOR_FDRP_TILE_WIDE_ROWS"):
        monkeypatch.delenv(k, raising=False)
    if request.param.startswith("wtile"):
        monkeypatch.setenv("METHEOR_FDRP_WTILE", "1")
        if request.param == "wtile_sub":
            monkeypatch.setenv("METHEOR_FDRP_WTILE_SUB", "1")
        if request.param == "wtile_heavy":
            monkeypatch.setenv("METHEOR_FDRP_WTILE_HEAVY", "1")
    elif request.param.startswith("walk") and len(request.param) > 4:
        monkeypatch.setenv("METHEOR_FDRP_WALK4", request.param[4:])
        monkeypatch.setenv("METHEOR_FDRP_TILE", "0")
    elif request.param.startswith("tile"):
        monkeypatch.setenv("METHEOR_FDRP_TILE", "1")
        if request.param == "tile_wide_rows":
            monkeypatch.setenv("METHEOR_FDRP_TILE_WIDE_ROWS", "1")
    elif request.param == "walk":
        monkeypatch.setenv("METHEOR_FDRP_TILE", "0")
        monkeypatch.setenv("METHEOR_FDRP_WALK4", "0")
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
            sub = shard.slice_region(c, b, e, halo=shard.max_span(c) + 202) if regions else c
            bt = util.device_batch(sub, region=(b, e), device=device)
            keep.append(bt)
            eng.fdrp_accumulate(bt, **kw)
    return eng.fdrp_fetch()


def same(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a - b)
    d[both_nan] = 0
    assert not np.isnan(d).any()
    return d.max() if len(d) else 0.0


def check(dev, reads, kw):
    of, oq = reads.fdrp(**kw), reads.qfdrp(**kw)
    assert len(dev["pos"]) == len(of) == len(oq)
    assert (dev["tid"] == of.tid).all() and (dev["pos"] == of.pos[:, 0]).all()
    assert (dev["n_reads"] == of.cnt[:, 0]).all()
    df, dq = same(dev["fdrp"], of.val), same(dev["qfdrp"], oq.val)
    assert df <= TOL and dq <= TOL, (df, dq)
    nf = int((dev["fdrp"].view(np.uint32) != of.val.view(np.uint32)).sum())
    nq = int((dev["qfdrp"].view(np.uint32) != oq.val.view(np.uint32)).sum())
    return len(of), nf, nq


def fixture(golden_dir, k):
    rec = bamio.read_bam(os.path.join(golden_dir, "test%d.bam" % k))
    reads = pyoracle.Reads.decode(rec)
    return reads, util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])


# ---- the reference's fixtures and known answers (fdrp.rs:252-332, qfdrp.rs:357-439) -----------------
def test_reference_fixtures(eng, golden_dir):
    kw0 = dict(min_qual=0, min_depth=2, max_depth=40, min_overlap=4)
    kw1 = dict(min_qual=1, min_depth=2, max_depth=40, min_overlap=4)
    reads, c = fixture(golden_dir, 1)
    d = run_device(eng, [c], kw0)
    assert d["pos"].tolist() == [0, 2, 4, 6] and (d["fdrp"] == 1.0).all()                     # fdrp.rs:253-268
    assert (np.abs(d["qfdrp"] - f32(8.0 / 15.0)) < 1e-5).all() and (d["n_reads"] == 16).all()  # qfdrp.rs:357-374, 309-332
    check(d, reads, kw0)
    reads, c = fixture(golden_dir, 2)
    d = run_device(eng, [c], kw0)
    assert (np.abs(d["fdrp"] - (1.0 - 56.0 / 120.0)) < 1e-4).all()                            # fdrp.rs:270-285
    assert (d["qfdrp"] == f32(8.0 / 15.0)).all()                                               # qfdrp.rs:376-392 (exact)
    check(d, reads, kw0)
    reads, c = fixture(golden_dir, 3)
    d = run_device(eng, [c], kw1)
    assert d["pos"].tolist() == [0, 2, 4, 6] and (d["fdrp"] == 1.0).all() and (d["qfdrp"] == 1.0).all()   # fdrp.rs:287-302, qfdrp.rs:394-409
    check(d, reads, kw1)
    reads, c = fixture(golden_dir, 4)
    d = run_device(eng, [c], kw1)
    assert d["pos"].tolist() == [0, 2, 4, 6, 13, 15, 17, 19] and (d["fdrp"] == 1.0).all()     # fdrp.rs:304-319
    assert (d["qfdrp"] == f32(8.0 / 15.0)).all()                                               # qfdrp.rs:411-427 (exact)
    check(d, reads, kw1)
    reads, c = fixture(golden_dir, 5)
    assert len(run_device(eng, [c], kw1)["pos"]) == 0                                          # fdrp.rs:321-332
    # defaults on test1 -> 4 sites with value 0 (8-bp reads never reach --min-overlap 35)
    reads, c = fixture(golden_dir, 1)
    d = run_device(eng, [c], dict())
    assert d["pos"].tolist() == [0, 2, 4, 6] and (d["fdrp"] == 0).all() and (d["qfdrp"] == 0).all()
    check(d, reads, dict())


def test_call_outside_covered_interval(eng):
    """a reverse read reports its first CpG at start-1: bit1 (call) without bit0 (covered).  Such a
    position counts in qFDRP's num_overlap_cpgs (qfdrp.rs:115) but never as a difference (fdrp.rs:114)."""
    from metheor_amd import synth
    rows = [(1000, 1, [(0, 1000, 1), (10, 1010, 1)]), (1001, 0, [(0, 1000, 0), (9, 1010, 1)]), (1001, 0, [(0, 1000, 0), (9, 1010, 0)])]
    start = np.array([r[0] for r in rows], np.int32)
    off = np.array([0, 2, 4, 6], np.uint32)
    pos = np.array([p | (m << 31) for r in rows for (_, p, m) in r[2]], np.uint32)
    rel = np.array([q for r in rows for (q, _, _) in r[2]], np.uint8)
    c = dict(tid=0, length=10_000, read_start=start, read_end=start + 49, read_mapq=np.full(3, 40, np.uint8),
             read_fwd=np.array([r[1] for r in rows], np.uint8), cpg_off=off, cpg_pos=pos, cpg_rel=rel)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_qual=0, min_depth=0, max_depth=40, min_overlap=1)
    d = run_device(eng, [c], kw)
    check(d, reads, kw)
    row = {int(p): (float(a), float(b), int(n)) for p, a, b, n in zip(d["pos"], d["fdrp"], d["qfdrp"], d["n_reads"])}
    # site 1000: pairs (0,1): 1000 uncovered by read 1 -> only 1010 compared (equal) -> ham 0/2; (0,2): 1010 differs -> 1/2; (1,2): 1/2
    assert row[1000][2] == 3 and abs(row[1000][0] - 2.0 / 3.0) < 1e-6 and abs(row[1000][1] - (0.5 + 0.5) / 3.0) < 1e-6


def test_real_rrbs_reads(eng, golden_dir):
    rec = bamio.read_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    reads = pyoracle.Reads.decode(rec)
    c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
    for kw in (dict(min_qual=10, min_depth=10, max_depth=40, min_overlap=35), dict(min_qual=10, min_depth=2, max_depth=64, min_overlap=10),
               dict(min_qual=43, min_depth=0, max_depth=40, min_overlap=0), dict(min_qual=0, min_depth=3, max_depth=5, min_overlap=20, seed=7)):
        print("RRBS", kw, "rows/notbit(f)/notbit(q):", check(run_device(eng, [c], kw), reads, kw))


@pytest.mark.parametrize("device_mem", [False, True])
def test_synthetic_vs_oracle(eng, device_mem):
    from metheor_amd import synth
    c = synth.make_contig(1, 600_000, 100_000, 0.02, np.random.default_rng(61))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    dev = "cuda:0" if device_mem else None
    for kw in (dict(min_qual=10, min_depth=10, max_depth=64, min_overlap=35),     # depth ~25x <= 64: no sampling, exact
               dict(min_qual=0, min_depth=0, max_depth=64, min_overlap=0),
               dict(min_qual=10, min_depth=10, max_depth=12, min_overlap=35, seed=1234)):   # reservoir branch (shared draw)
        n, nf, nq = check(run_device(eng, [c], kw, device=dev), reads, kw)
        print("synthetic", kw, "rows", n, "not bit-identical fdrp/qfdrp:", nf, nq)
        assert n > 5000


def test_hotspot_depth_50(eng):
    """BASELINE config 4 in small: 1-kbp windows at exactly 50x, -D 64 (no sampling => exact parity)"""
    from metheor_amd import synth
    c = synth.hotspots(n_windows=40, window=1000, depth=50, density=0.08, seed=50)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_qual=10, min_depth=10, max_depth=64, min_overlap=35)
    n, nf, nq = check(run_device(eng, [c], kw), reads, kw)
    print("hotspot rows", n, "not bit-identical fdrp/qfdrp:", nf, nq)
    assert n > 1000


def test_window_drop_and_nan_rows(eng):
    """400-bp forward reads mostly do not fit the +-201-bp array (fdrp.rs:58-63): they still create the
    site's entry, so with min_depth 0 the reference emits 0/0 = NaN rows (release arithmetic)"""
    from metheor_amd import synth
    c = synth.make_contig(0, 120_000, 6_000, 0.02, np.random.default_rng(63), read_len=400)
    c = util.subset_reads(c, c["read_fwd"] == 1)        # a reverse read's call at start-1 can index -1: the reference panics there
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_qual=10, min_depth=0, max_depth=40, min_overlap=35)
    d = run_device(eng, [c], kw)
    n, nf, nq = check(d, reads, kw)
    assert np.isnan(d["fdrp"]).sum() > 100 and (d["n_reads"] <= 1)[np.isnan(d["fdrp"])].all()
    kw = dict(min_qual=10, min_depth=2, max_depth=40, min_overlap=35)
    check(run_device(eng, [c], kw), reads, kw)


def test_dense_window_takes_call_by_call_path(eng):
    """a CpG every ~4 bp: > 64 sites inside +-200 bp (no 64-bit site masks) and ~37 calls per read (more than
    the 16 call registers of a stored read) -- the kernel's call-by-call pair evaluation"""
    from metheor_amd import synth
    c = synth.make_contig(0, 60_000, 8_000, 0.25, np.random.default_rng(67))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    assert len(c["cpg_pos"]) / len(c["read_start"]) > 20
    for kw in (dict(min_qual=10, min_depth=5, max_depth=64, min_overlap=35), dict(min_qual=0, min_depth=0, max_depth=10, min_overlap=0, seed=3)):
        n, nf, nq = check(run_device(eng, [c], kw), reads, kw)
        print("dense", kw, "rows", n, "not bit-identical fdrp/qfdrp:", nf, nq)
        assert n > 5000


def test_flush_then_reopen_inside_one_site(eng):
    """fdrp.rs:212-228: a read that starts at/before c+1 but whose first call lies past c flushes site c; a later
    read calling c re-opens it and the LAST segment is what the reference reports (HashMap insert overwrites)"""
    from metheor_amd import synth
    # (start, fwd, [(rel, pos, meth)])
    rows = [(1000, 1, [(0, 1000, 1), (50, 1050, 1)]),      # opens 1000 and 1050
            (1000, 1, [(0, 1000, 0), (50, 1050, 0)]),
            (1001, 1, [(49, 1050, 1)]),                     # first call 1050 > 1000: flushes site 1000 (2 reads)
            (1001, 0, [(0, 1000, 1), (49, 1050, 0)]),      # reverse read calling start-1 = 1000: re-opens site 1000
            (1001, 0, [(0, 1000, 0), (49, 1050, 1)]),
            (1001, 0, [(0, 1000, 0), (49, 1050, 1)]),
            (1500, 1, [(0, 1500, 1)])]
    start = np.array([r[0] for r in rows], np.int32)
    off = np.cumsum([0] + [len(r[2]) for r in rows]).astype(np.uint32)
    pos = np.array([p | (m << 31) for r in rows for (_, p, m) in r[2]], np.uint32)
    rel = np.array([q for r in rows for (q, _, _) in r[2]], np.uint8)
    c = dict(tid=0, length=10_000, read_start=start, read_end=start + 59, read_mapq=np.full(len(rows), 40, np.uint8),
             read_fwd=np.array([r[1] for r in rows], np.uint8), cpg_off=off, cpg_pos=pos, cpg_rel=rel)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    for kw in (dict(min_qual=0, min_depth=0, max_depth=40, min_overlap=1), dict(min_qual=0, min_depth=3, max_depth=40, min_overlap=1),
               dict(min_qual=0, min_depth=2, max_depth=2, min_overlap=1, seed=5)):
        d = run_device(eng, [c], kw)
        check(d, reads, kw)
    kw = dict(min_qual=0, min_depth=0, max_depth=40, min_overlap=1)
    d = run_device(eng, [c], kw)
    row = {int(p): int(n) for p, n in zip(d["pos"], d["n_reads"])}
    assert row[1000] == 3 and row[1050] == 6, row      # site 1000: the re-opened segment (3 reverse reads), not the first 2


def test_multi_contig_and_region_split(eng):
    from metheor_amd import shard, synth
    rng = np.random.default_rng(65)
    cs = [synth.make_contig(0, 150_000, 25_000, 0.03, rng), synth.make_contig(1, 300_000, 55_000, 0.03, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    kw = dict(min_qual=10, min_depth=10, max_depth=64, min_overlap=35)
    d = run_device(eng, cs, kw, regions=[[(0, cs[0]["length"])], shard.plan_regions(cs[1], 4)])
    check(d, reads, kw)
    d2 = run_device(eng, cs, kw)
    assert (d["pos"] == d2["pos"]).all() and (d["qfdrp"].view(np.uint32) == d2["qfdrp"].view(np.uint32)).all()


def test_capacity_and_empty(eng):
    from metheor_amd import Batch, MthError
    z4 = np.zeros(0, np.int32)
    b = Batch(0, 0, 1000, z4, z4, np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    eng.reset()
    eng.fdrp_accumulate(b)
    assert len(eng.fdrp_fetch()["pos"]) == 0
    eng.fdrp_accumulate(b, max_depth=65)           # accepted: the limit is 64 reads STORED for a site (test_max_depth_above_64)
    assert len(eng.fdrp_fetch()["pos"]) == 0
    eng.reset()


def test_max_depth_above_64(eng, walk_choice):
    """the main pass stores at most 64 reads per site (one lane each); with max_depth > 64 deeper sites are redone by a
    256-slot pass (pair-parallel merge of the call lists) -- exact, including the reservoir branch beyond max_depth;
    sites beyond 256 stored reads by a third pass with rows in HBM scratch (round 2: the reference has no such limit;
    output_validation.rs runs --max-depth 100); only max_depth > 16384 is refused"""
    from metheor_amd import MthError, synth
    if walk_choice in ("walk16", "walk32", "wtile_sub", "wtile_heavy"):
        pytest.skip("40 s a form: the deep passes sit behind the first pass whatever its form -- host's choice, tile and walk cover them")
    c = synth.make_contig(0, 300_000, 50_000, 0.03, np.random.default_rng(91))           # ~25x: every site below 64 reads
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_qual=10, min_depth=5, max_depth=100, min_overlap=20)
    check(run_device(eng, [c], kw), reads, kw)
    deep = synth.make_contig(0, 20_000, 12_000, 0.03, np.random.default_rng(92))         # ~90x: sites with 64..130 reads
    dreads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(deep))
    for kw in (dict(min_qual=10, min_depth=5, max_depth=100, min_overlap=20, seed=3),     # stored in full up to 100, sampled beyond
               dict(min_qual=0, min_depth=5, max_depth=200, min_overlap=1),               # never sampled
               dict(min_qual=10, min_depth=5, max_depth=64, min_overlap=20, seed=5)):     # the 64-slot pass alone, sampling
        n, nf, nq = check(run_device(eng, [deep], kw), dreads, kw)
        assert n > 400
    assert int(dreads.fdrp(min_qual=0, min_depth=5, max_depth=200, min_overlap=1).cnt[:, 0].max()) > 64
    # region split + device-resident batches through the deep pass
    from metheor_amd import shard
    kw = dict(min_qual=10, min_depth=5, max_depth=150, min_overlap=20)
    check(run_device(eng, [deep], kw, device="cuda:0", regions=[shard.plan_regions(deep, 3)]), dreads, kw)
    very = synth.make_contig(0, 6_000, 14_000, 0.03, np.random.default_rng(93))          # ~350x: more than 256 reads on a site
    vreads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(very))
    kw = dict(min_qual=10, min_depth=5, max_depth=256, min_overlap=20, seed=9)            # 256 slots hold it, sampling beyond
    check(run_device(eng, [very], kw), vreads, kw)
    # beyond 256 stored reads: the third pass (rows in HBM scratch) -- sampled at 300, stored in full at 1000 and 16384
    for kw in (dict(min_qual=10, min_depth=5, max_depth=300, min_overlap=20, seed=4), dict(min_qual=10, min_depth=5, max_depth=1000, min_overlap=20),
               dict(min_qual=0, min_depth=300, max_depth=16384, min_overlap=1)):
        n, nf, nq = check(run_device(eng, [very], kw), vreads, kw)
        assert n > 50
    assert int(vreads.fdrp(min_qual=10, min_depth=5, max_depth=1000, min_overlap=20).cnt[:, 0].max()) > 256
    check(run_device(eng, [very], dict(min_qual=10, min_depth=5, max_depth=400, min_overlap=20, seed=2), device="cuda:0",
                     regions=[shard.plan_regions(very, 2)]), vreads, dict(min_qual=10, min_depth=5, max_depth=400, min_overlap=20, seed=2))
    with pytest.raises(MthError) as e:
        run_device(eng, [very], dict(min_qual=10, min_depth=5, max_depth=16385, min_overlap=20))
    assert e.value.status == -8
    eng.reset()


def test_wgbs_depth(eng):
    """BASELINE config 3 in small (density 0.0091, ~10x, a dozen candidate reads per site): bit for bit the oracle's fdrp / qfdrp
    columns, including max_depth below the depth (reservoir, shared draw) and min_depth 1 (one-read sites: 0 / 0 = NaN rows)"""
    from metheor_amd import synth
    c = synth.make_contig(2, 3_000_000, 200_000, 0.0091, np.random.default_rng(77))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    for kw in (dict(min_qual=10, min_depth=10, max_depth=40, min_overlap=35),
               dict(min_qual=10, min_depth=4, max_depth=40, min_overlap=35),
               dict(min_qual=0, min_depth=1, max_depth=64, min_overlap=0),
               dict(min_qual=10, min_depth=2, max_depth=5, min_overlap=20, seed=99)):
        n, nf, nq = check(run_device(eng, [c], kw, device="cuda:0"), reads, kw)
        print("wgbs depth", kw, "rows", n, "not bit-identical:", nf, nq)
        assert n > 10000 and nf == 0 and nq == 0


# ---- the reservoir as a reservoir (the draw itself: tests/test_reservoir.py) -----------------------------
def test_reservoir_set_size_and_unbiased_estimates(eng):
    """BASELINE config 4's hotspots with the CLI's DEFAULT -D 40 (the sampling branch at 50x): every site stores min(depth, 40)
    reads; FDRP and qFDRP of a uniformly drawn 40-subset are unbiased estimates of the full-depth values (a pair is kept with
    the same probability whatever its two reads), so over thousands of sampled sites the mean difference to -D 64 must vanish
    within its standard error -- a reservoir that favoured early (= leftmost) reads would not: neighbours in coordinate order
    overlap more, fewer of their pairs fall under --min-overlap.  Bound stated in profiles/HISTORY.md section 12: |z| < 5."""
    from metheor_amd import synth
    c = synth.hotspots(n_windows=500, window=1000, depth=50, density=0.08, seed=51)
    full = run_device(eng, [c], dict(min_qual=10, min_depth=10, max_depth=64, min_overlap=35, seed=1))
    seeds = {}
    for seed in (1, 2, 3):
        d = run_device(eng, [c], dict(min_qual=10, min_depth=10, max_depth=40, min_overlap=35, seed=seed))
        assert (d["pos"] == full["pos"]).all()
        assert (d["n_reads"] == np.minimum(full["n_reads"], 40)).all()          # stored set size = min(depth, D)
        seeds[seed] = d
    deep = full["n_reads"] > 40
    assert deep.sum() > 5000
    # the draw depends on the seed: sampled sites change, unsampled ones cannot
    assert (seeds[1]["fdrp"][~deep].view(np.uint32) == full["fdrp"][~deep].view(np.uint32)).all()
    assert (seeds[1]["qfdrp"][deep] != seeds[2]["qfdrp"][deep]).mean() > 0.5
    for key in ("fdrp", "qfdrp"):
        zs = []
        for seed, d in seeds.items():
            diff = d[key][deep].astype(np.float64) - full[key][deep].astype(np.float64)
            z = diff.mean() / (diff.std(ddof=1) / np.sqrt(len(diff)))
            zs.append(z)
            assert abs(z) < 5.0, (key, seed, z, diff.mean())
        print(key, "z-scores of mean(-D 40 minus -D 64) over", int(deep.sum()), "sites:", [round(z, 2) for z in zs])
    # what a NON-uniform reservoir looks like to this test: keep each site's first 40 reads (coordinate order) -- the oracle on a
    # copy of the input from which every site's later reads are gone is not needed: dropping the LAST 10 reads of every window
    # shifts the mean well outside the bound
    starts = c["read_start"].astype(np.int64)
    stride = 1000 + 2 * 150 + 404
    win = starts // stride
    order_in_win = np.arange(len(starts)) - np.searchsorted(win, win, side="left")
    per = np.bincount(win)[win]
    biased = util.subset_reads(c, order_in_win < per * 0.8)
    b = run_device(eng, [biased], dict(min_qual=10, min_depth=10, max_depth=64, min_overlap=35, seed=1))
    common, ia, ib = np.intersect1d(full["pos"][deep], b["pos"], return_indices=True)
    diff = b["fdrp"][ib].astype(np.float64) - full["fdrp"][deep][ia].astype(np.float64)
    zb = diff.mean() / (diff.std(ddof=1) / np.sqrt(len(diff)))
    print("control (first 80 % of every window's reads instead of a uniform subset): z =", round(zb, 1))
    assert abs(zb) > 8.0
