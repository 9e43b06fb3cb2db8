"""GPU parity: the fused PDR+LPMD HIP path (through the C ABI) against the CPU oracle.

Bar: n_concordant / n_discordant / LPMD counters bit-exact; PDR and LPMD floats bit-exact (they
are computed from the integers with the reference's f32 expressions, pdr.rs:47-49, lpmd.rs:51-55).
"""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu

f32 = np.float32


@pytest.fixture(autouse=True, params=["auto", "tile", "runs", "wide14", "wide15", "wide16"])
def kernel_form(request, monkeypatch):
    """every case runs with the engine's own choice of kernel form (MTH_PDR_WIDE unset: the density chooser of
    launch_pdr_lpmd, the path production takes), through the dense tile kernel (forced: MTH_PDR_WIDE=0), through the
    persistent run form of the dense kernel (round 5: k_pdr_lpmd_runs + k_gather_runs, MTH_TILE_RUNS=1) and through the
    hashed-site form for sparse batches (mth_pdr_wide.hip) with 16384-, 32768- and 65536-bp tiles forced"""
    monkeypatch.delenv("MTH_PDR_WIDE", raising=False)
    monkeypatch.delenv("MTH_TILE_RUNS", raising=False)
    if request.param == "tile":
        monkeypatch.setenv("MTH_PDR_WIDE", "0")
    elif request.param == "runs":
        monkeypatch.setenv("MTH_PDR_WIDE", "0")
        monkeypatch.setenv("MTH_TILE_RUNS", "1")
    elif request.param.startswith("wide"):
        monkeypatch.setenv("MTH_PDR_WIDE", request.param[4:])
    return request.param


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


def run_device(eng, contigs, params, device=None, regions=None, rel16=False):
    """contigs: list of contig dicts (tid-sorted); regions: optional per-contig list of (beg,end)"""
    from metheor_amd import shard
    eng.reset()
    keep = []
    for ci, c in enumerate(contigs):
        regs = regions[ci] if regions else [(0, c["length"])]
        for (b, e) in regs:
            sub = shard.slice_region(c, b, e) if regions else c
            bt = util.device_batch(sub, region=(b, e), device=device, rel16=rel16)
            keep.append(bt)
            eng.pdr_lpmd_accumulate(bt, params)
    return eng.pdr_fetch(), eng.lpmd_global()


def check_against_oracle(dev_pdr, dev_lpmd, reads, pdr_kw, lpmd_kw):
    o = reads.pdr(**pdr_kw)
    assert len(dev_pdr["pos"]) == len(o), (len(dev_pdr["pos"]), len(o))
    assert (dev_pdr["tid"] == o.tid).all()
    assert (dev_pdr["pos"] == o.pos[:, 0]).all()
    assert (dev_pdr["n_concordant"] == o.cnt[:, 0]).all()
    assert (dev_pdr["n_discordant"] == o.cnt[:, 1]).all()
    assert (dev_pdr["pdr"].view(np.uint32) == o.val.view(np.uint32)).all()   # bit-exact f32
    l = reads.lpmd(**lpmd_kw)
    for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"):
        assert dev_lpmd[k] == l[k], (k, dev_lpmd[k], l[k])
    a, b = f32(dev_lpmd["lpmd"]), f32(l["lpmd"])
    assert (np.isnan(a) and np.isnan(b)) or a.view(np.uint32) == b.view(np.uint32)


def fixture_contig(golden_dir, k):
    rec = bamio.read_bam(os.path.join(golden_dir, "test%d.bam" % k))
    reads = pyoracle.Reads.decode(rec)
    return reads, util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])


# ---- the reference's own fixtures and known answers (pdr.rs:218-366, lpmd.rs:208-269) ----------
@pytest.mark.parametrize("k,nsite,pdr,nc,nd,lpmd", [
    (1, 4, 0.875, 2, 14, 0.5), (2, 4, 0.0, 16, 0, 0.0), (3, 4, 0.0, 2, 0, 0.0),
    (4, 8, 0.875, 2, 14, 0.5), (5, 0, None, None, None, float("nan"))])
def test_reference_fixtures(eng, golden_dir, k, nsite, pdr, nc, nd, lpmd):
    from metheor_amd import PdrLpmdParams
    reads, c = fixture_contig(golden_dir, k)
    # whole chr1 (249 Mbp) as one region is what a drop-in host would submit
    p, l = run_device(eng, [c], PdrLpmdParams(min_depth=0, min_cpgs=0, min_qual=10))
    assert len(p["pos"]) == nsite
    if nsite:
        assert (p["pdr"] == f32(pdr)).all() and (p["n_concordant"] == nc).all() and (p["n_discordant"] == nd).all()
    assert (np.isnan(lpmd) and np.isnan(l["lpmd"])) or l["lpmd"] == f32(lpmd)
    check_against_oracle(p, l, reads, dict(min_depth=0, min_cpgs=0, min_qual=10), dict())


def test_reference_fixture6_min_cpgs(eng, golden_dir):   # pdr.rs:326-365
    from metheor_amd import PdrLpmdParams
    reads, c = fixture_contig(golden_dir, 6)
    p, l = run_device(eng, [c], PdrLpmdParams(min_depth=0, min_cpgs=1))
    assert p["pos"].tolist() == [2, 13] and (p["n_concordant"] == 16).all() and (p["pdr"] == 0).all()
    p, l = run_device(eng, [c], PdrLpmdParams(min_depth=0, min_cpgs=2))
    assert len(p["pos"]) == 0


def test_default_cli_golden(eng, golden_dir):   # metheor pdr -i tests/test1.bam defaults
    from metheor_amd import PdrLpmdParams
    reads, c = fixture_contig(golden_dir, 1)
    p, l = run_device(eng, [c], PdrLpmdParams())
    rows = list(zip(p["pos"].tolist(), p["pdr"].tolist(), p["n_concordant"].tolist(), p["n_discordant"].tolist()))
    assert rows == [(0, 0.875, 2, 14), (2, 0.875, 2, 14), (4, 0.875, 2, 14), (6, 0.875, 2, 14)]
    assert l["lpmd"] == f32(0.5)


# ---- real reads (the reference's 1000-read RRBS SAM fixture: reverse strand, 25-29 bp) -----------
def test_real_rrbs_reads(eng, golden_dir):
    from metheor_amd import PdrLpmdParams
    rec = bamio.read_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    reads = pyoracle.Reads.decode(rec)
    c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
    for kw in (dict(min_depth=0, min_cpgs=0, min_qual=10), dict(min_depth=10, min_cpgs=4, min_qual=10),
               dict(min_depth=3, min_cpgs=2, min_qual=43)):
        p, l = run_device(eng, [c], PdrLpmdParams(**kw))
        check_against_oracle(p, l, reads, kw, dict())
    for lk in (dict(min_distance=1, max_distance=3), dict(min_distance=5, max_distance=4),
               dict(min_distance=0, max_distance=200), dict(min_distance=2, max_distance=16, min_qual=50)):
        pr = PdrLpmdParams(min_distance=lk["min_distance"], max_distance=lk["max_distance"],
                           lpmd_min_qual=lk.get("min_qual", 10))
        p, l = run_device(eng, [c], pr)
        check_against_oracle(p, l, reads, dict(), lk)


# ---- seeded synthetic WGBS vs oracle -----------------------------------------------------------
@pytest.mark.parametrize("device_mem", [False, True])
def test_synthetic_vs_oracle(eng, device_mem):
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(7)
    c = synth.make_contig(3, 2_000_000, 300_000, 0.02, rng)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    dev = "cuda:0" if device_mem else None
    for kw in (dict(min_depth=10, min_cpgs=4, min_qual=10), dict(min_depth=0, min_cpgs=0, min_qual=0)):
        p, l = run_device(eng, [c], PdrLpmdParams(**kw), device=dev)
        check_against_oracle(p, l, reads, kw, dict())
        assert len(p["pos"]) > 1000


def test_multi_contig_and_region_split(eng):
    """3 contigs; the middle one submitted as 5 region batches with halo reads: identical rows"""
    from metheor_amd import PdrLpmdParams, shard, synth
    rng = np.random.default_rng(11)
    cs = [synth.make_contig(t, ln, n, 0.03, rng) for t, (ln, n) in enumerate([(300_000, 40_000), (1_000_000, 150_000), (50_000, 3_000)])]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    regions = [[(0, cs[0]["length"])], shard.plan_regions(cs[1], 5), [(0, cs[2]["length"])]]
    kw = dict(min_depth=5, min_cpgs=2, min_qual=10)
    p, l = run_device(eng, cs, PdrLpmdParams(**kw), regions=regions)
    check_against_oracle(p, l, reads, kw, dict())
    p2, l2 = run_device(eng, cs, PdrLpmdParams(**kw))
    for k in p:
        assert (p[k] == p2[k]).all()
    assert l == l2 or all((l[k] == l2[k]) or (np.isnan(l[k]) and np.isnan(l2[k])) for k in l)


def test_high_depth_site_and_u16_rel(eng):
    """20 000 identical reads on one spot (LDS atomic contention, counters > u16) + u16 relpos"""
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(5)
    starts = np.sort(np.concatenate([np.full(20_000, 5000), rng.integers(0, 20_000, 2000)])).astype(np.int32)
    c = synth.make_contig(0, 30_000, len(starts), 0.05, rng, starts=starts)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_depth=10, min_cpgs=4, min_qual=10)
    for rel16 in (False, True):
        p, l = run_device(eng, [c], PdrLpmdParams(**kw), rel16=rel16)
        check_against_oracle(p, l, reads, kw, dict())
    assert p["n_concordant"].max() + p["n_discordant"].max() > 10_000


def test_deep_amplicon_wide_counters(eng):
    """The tile keeps ONE 32-bit LDS word per position (two 16-bit counts) while it has <= 65535 candidate reads;
    heavier tiles take two passes with 32-bit counters.  70 000 reads calling the same four CpGs (a deep
    amplicon: per-site counts beyond 16 bits) plus a uniformly deep tile with > 65535 candidates."""
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(81)
    n = 70_000
    start = np.sort(1000 + (np.arange(n) % 2)).astype(np.int32)
    cpgs = np.array([1005, 1015, 1040, 1080], np.int32)
    meth = rng.random((n, 4)) < np.where(rng.random((n, 1)) < 0.5, 0.9, 0.1)      # mostly concordant reads, some mixed
    pos = (cpgs[None, :].astype(np.uint32) | (meth.astype(np.uint32) << 31)).reshape(-1)
    rel = (cpgs[None, :] - start[:, None]).astype(np.uint8).reshape(-1)
    amp = dict(tid=0, length=9_000, read_start=start, read_end=start + 99, read_mapq=np.full(n, 40, np.uint8),
               read_fwd=np.ones(n, np.uint8), cpg_off=(np.arange(n + 1) * 4).astype(np.uint32), cpg_pos=pos, cpg_rel=rel)
    deep = synth.make_contig(1, 7_000, 75_000, 0.04, rng)                          # ~1600x over 7 kbp: two heavy tiles
    cs = [amp, deep]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    p = PdrLpmdParams(min_depth=10, min_cpgs=2, min_qual=10, min_distance=2, max_distance=40, lpmd_min_qual=10)
    d, l = run_device(eng, cs, p)
    check_against_oracle(d, l, reads, dict(min_depth=10, min_cpgs=2, min_qual=10), dict(min_distance=2, max_distance=40, min_qual=10))
    amp_rows = d["tid"] == 0
    assert amp_rows.sum() == 4 and ((d["n_concordant"] + d["n_discordant"])[amp_rows] == n).all()   # > 65535 per site


def test_edge_cases(eng):
    from metheor_amd import Batch, MthError, PdrLpmdParams, synth
    z4 = np.zeros(0, np.int32)
    # empty batch
    eng.reset()
    b = Batch(0, 0, 1000, z4, z4, np.zeros(0, np.uint8), np.zeros(1, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    eng.pdr_lpmd_accumulate(b, PdrLpmdParams())
    assert eng.pdr_count() == 0
    g = eng.lpmd_global()
    assert g["n_read"] == 0 and np.isnan(g["lpmd"])
    # reads without any CpG call
    eng.reset()
    st = np.array([5, 10, 10, 700], np.int32)
    b = Batch(0, 0, 1000, st, st + 149, np.full(4, 42, np.uint8), np.zeros(5, np.uint32), np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    eng.pdr_lpmd_accumulate(b, PdrLpmdParams(min_depth=0, min_cpgs=0))
    assert eng.pdr_count() == 0 and eng.lpmd_global()["n_valid_read"] == 4
    # unsorted reads are refused loudly
    rng = np.random.default_rng(1)
    c = synth.make_contig(0, 100_000, 5000, 0.02, rng)
    bad = dict(c)
    bad["read_start"] = c["read_start"][::-1].copy()
    bad["read_end"] = c["read_end"][::-1].copy()
    eng.reset()
    eng.pdr_lpmd_accumulate(util.device_batch(bad), PdrLpmdParams())
    with pytest.raises(MthError) as e:
        eng.pdr_count()
    assert e.value.status == -4
    # a lying max_span is detected
    eng.reset()
    bt = util.device_batch(c)
    bt.c.max_span = 100
    eng.pdr_lpmd_accumulate(bt, PdrLpmdParams())
    with pytest.raises(MthError) as e:
        eng.pdr_count()
    assert e.value.status == -5
    eng.reset()


# ---- spans beyond the 150-bp flush margin: the exact site walk (flush / re-open, pdr.rs:160-177) ------
def test_pdr_reopen_hand_built(eng):
    """read B (passing, span 400) starts before site c=1000 but its FIRST CpG is at 1200 > c+150: it
    flushes c; read C then contributes to c again and re-opens it.  Last segment with coverage >=
    min_depth wins; a shallow later segment does not erase an earlier qualifying one."""
    from metheor_amd import PdrLpmdParams, synth
    def contig(rows):
        start = np.array([r[0] for r in rows], np.int32)
        end = np.array([r[1] for r in rows], np.int32)
        off = np.zeros(len(rows) + 1, np.uint32)
        pos, rel = [], []
        for i, r in enumerate(rows):
            for (p, m) in r[2]:
                pos.append(p | (int(m) << 31)); rel.append(p - r[0])
            off[i + 1] = len(pos)
        return dict(tid=0, length=50_000, read_start=start, read_end=end, read_mapq=np.full(len(rows), 40, np.uint8),
                    read_fwd=np.ones(len(rows), np.uint8), cpg_off=off, cpg_pos=np.array(pos, np.uint32),
                    cpg_rel=np.array(rel, np.uint16))
    rows = [(990, 1089, [(1000, 1), (1010, 1)])] * 3                  # segment 1: 3 concordant reads on c=1000
    rows += [(995, 1394, [(1200, 1), (1300, 0)])]                      # B: first CpG 1200 > 1000+150 -> flush
    rows += [(999, 1098, [(1000, 1), (1010, 0)])] * 2                  # segment 2: 2 discordant reads re-open c
    c = contig(rows)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    for md in (0, 2, 3):
        kw = dict(min_depth=md, min_cpgs=0, min_qual=10)
        p, l = run_device(eng, [c], PdrLpmdParams(**kw))
        check_against_oracle(p, l, reads, kw, dict())
        row = {int(q): (int(a), int(b)) for q, a, b in zip(p["pos"], p["n_concordant"], p["n_discordant"])}
        assert row[1000] == ((0, 2) if md <= 2 else (3, 0))


def test_pdr_long_spans_vs_oracle(eng):
    """400-bp reads (span > 150): sites are flushed and re-opened in the reference's stream; the device
    takes the site-walk path for PDR and the tile kernel for LPMD, in one accumulate call"""
    from metheor_amd import PdrLpmdParams, shard, synth
    rng = np.random.default_rng(51)
    cs = [synth.make_contig(0, 400_000, 30_000, 0.02, rng, read_len=400), synth.make_contig(1, 200_000, 9_000, 0.03, rng, read_len=400)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    nrows = []
    for kw in (dict(min_depth=10, min_cpgs=4, min_qual=10), dict(min_depth=0, min_cpgs=0, min_qual=0), dict(min_depth=40, min_cpgs=8, min_qual=10)):
        p, l = run_device(eng, cs, PdrLpmdParams(**kw))
        check_against_oracle(p, l, reads, kw, dict())
        nrows.append(len(p["pos"]))
    assert nrows[0] > 1000 and nrows[1] > nrows[0] and nrows[2] > 0      # not vacuous
    # the stream really re-opens sites here: plain pooled counting differs from the oracle somewhere
    kw = dict(min_depth=0, min_cpgs=0, min_qual=0)
    o = reads.pdr(**kw)
    pooled = np.bincount((cs[0]["cpg_pos"] & 0x7fffffff).astype(np.int64), minlength=cs[0]["length"])
    sel = o.tid == 0
    assert (pooled[o.pos[sel, 0]] != o.cnt[sel].sum(1)).sum() > 0
    # region-split batches of the long-span contig give the same rows; PDR-only and LPMD-only too
    regions = [shard.plan_regions(cs[0], 3), [(0, cs[1]["length"])]]
    kw = dict(min_depth=10, min_cpgs=4, min_qual=10)
    p2, l2 = run_device(eng, cs, PdrLpmdParams(**kw), regions=regions)
    check_against_oracle(p2, l2, reads, kw, dict())
    p3, _ = run_device(eng, cs, PdrLpmdParams(want_lpmd=False, **kw))
    assert (p3["pos"] == p2["pos"]).all() and (p3["n_discordant"] == p2["n_discordant"]).all()


@pytest.mark.parametrize("read_len", [120, 200, 255, 256, 257])
def test_counter_margin_boundary_spans(eng, read_len):
    """the tile kernel keeps 256 margin words on either side of a tile for batches with max_span <= 256 (no clamp per
    call slot) and falls back to the clamped form above that: spans on both sides of the switch, whole contigs and region
    splits (halo reads call positions on both sides of a region), PDR via the tile kernel (<= 150) or the exact walk fed by
    the tile kernel's site discovery (> 150), LPMD always via the tile kernel"""
    from metheor_amd import PdrLpmdParams, shard, synth
    rng = np.random.default_rng(9000 + read_len)
    cs = [synth.make_contig(0, 150_000, 24_000, 0.03, rng, read_len=read_len), synth.make_contig(1, 9_000, 900, 0.05, rng, read_len=read_len)]
    assert all(int((c["read_end"] - c["read_start"]).max()) + 1 == read_len for c in cs)
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    for kw, lkw in ((dict(min_depth=10, min_cpgs=4, min_qual=10), dict()), (dict(min_depth=0, min_cpgs=0, min_qual=0), dict(min_distance=1, max_distance=40, min_qual=0)),
                    (dict(min_depth=3, min_cpgs=1, min_qual=20), dict(min_distance=2, max_distance=300, min_qual=20))):
        params = PdrLpmdParams(lpmd_min_qual=lkw.get("min_qual", 10), min_distance=lkw.get("min_distance", 2), max_distance=lkw.get("max_distance", 16), **kw)
        p, l = run_device(eng, cs, params)
        check_against_oracle(p, l, reads, kw, lkw)
        assert len(p["pos"]) > 100
        regions = [shard.plan_regions(cs[0], 5), [(0, 4097), (4097, cs[1]["length"])]]
        p2, l2 = run_device(eng, cs, params, regions=regions)
        check_against_oracle(p2, l2, reads, kw, lkw)


# ---- BASELINE config 2 at full size: size-independent properties ----------------------------------
def test_full_size_properties(eng):
    """10 M reads: device totals vs closed forms computed with numpy on the SoA"""
    from metheor_amd import PdrLpmdParams, synth
    c = synth.chr19_10m()
    n = len(c["read_start"])
    bt = util.device_batch(c, device="cuda:0")
    eng.reset()
    eng.pdr_lpmd_accumulate(bt, PdrLpmdParams(min_depth=0, min_cpgs=0, min_qual=10))
    p, l = eng.pdr_fetch(), eng.lpmd_global()
    off = c["cpg_off"].astype(np.int64)
    ncpg = np.diff(off)
    ok = (c["read_mapq"] >= 10) & (ncpg > 0)
    meth = (c["cpg_pos"] >> 31).astype(np.int64)
    msum = np.add.reduceat(np.concatenate([meth, [0]]), off[:-1])
    msum[ncpg == 0] = 0
    disc = (msum > 0) & (msum < ncpg)
    # every call of a passing read lands in exactly one site counter
    assert int(p["n_concordant"].sum()) == int(ncpg[ok & ~disc].sum())
    assert int(p["n_discordant"].sum()) == int(ncpg[ok & disc].sum())
    # sites are the distinct called positions of passing reads, strictly increasing
    called = np.unique((c["cpg_pos"] & 0x7fffffff)[np.repeat(ok, ncpg)])
    assert len(p["pos"]) == len(called) and (p["pos"] == called).all()
    assert l["n_read"] == n and l["n_valid_read"] == int((c["read_mapq"] >= 10).sum())
    # idempotence: a second pass over the same resident batch gives identical rows
    eng.reset()
    eng.pdr_lpmd_accumulate(bt, PdrLpmdParams(min_depth=0, min_cpgs=0, min_qual=10))
    p2, l2 = eng.pdr_fetch(), eng.lpmd_global()
    assert all((p[k] == p2[k]).all() for k in p) and l2["n_concordant"] == l["n_concordant"]
    # oracle on a 300k-read prefix region (exact), same resident arrays submitted as a sub-region
    from metheor_amd import shard
    cut = int(c["read_start"][300_000])
    sub = shard.slice_region(c, 0, cut)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(sub))
    o = reads.pdr(min_depth=0, min_cpgs=0, min_qual=10)
    keep = o.pos[:, 0] < cut
    m = p["pos"] < cut
    assert (p["pos"][m] == o.pos[keep, 0]).all()
    assert (p["n_concordant"][m] == o.cnt[keep, 0]).all() and (p["n_discordant"][m] == o.cnt[keep, 1]).all()
    assert (p["pdr"][m].view(np.uint32) == o.val[keep].view(np.uint32)).all()


# ---- paths of the wave-cooperative kernel that ordinary WGBS density never reaches -----------------
def test_dense_cpg_chunks(eng):
    """~30 calls/read: a 64-read chunk has > 1024 calls (chunk splitting) and > 8 call rounds
    (the part of the scatter walk that re-derives read ids), look-backs run deeper than 8 calls"""
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(21)
    c = synth.make_contig(0, 200_000, 40_000, 0.2, rng)
    assert c["cpg_off"][-1] / len(c["read_start"]) > 20
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    for kw, lk in ((dict(min_depth=10, min_cpgs=4, min_qual=10), dict()),
                   (dict(min_depth=0, min_cpgs=0, min_qual=0), dict(min_distance=1, max_distance=60))):
        pr = PdrLpmdParams(min_distance=lk.get("min_distance", 2), max_distance=lk.get("max_distance", 16), **kw)
        p, l = run_device(eng, [c], pr)
        check_against_oracle(p, l, reads, kw, lk)


def test_long_reads_lpmd_only(eng):
    """6-kbp reads with > 1024 calls each (16-bit relpos): the LPMD-only launch of the tile kernel and its
    memory loop for reads with more calls than the 8 register slots"""
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(22)
    c = synth.make_contig(0, 400_000, 600, 0.3, rng, read_len=6000)
    assert c["cpg_rel"].dtype == np.uint16 and np.diff(c["cpg_off"].astype(np.int64)).max() > 1024
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    for lk in (dict(min_distance=2, max_distance=16), dict(min_distance=0, max_distance=300)):
        pr = PdrLpmdParams(min_distance=lk["min_distance"], max_distance=lk["max_distance"], want_pdr=False)
        eng.reset()
        eng.pdr_lpmd_accumulate(util.device_batch(c), pr)
        l, o = eng.lpmd_global(), reads.lpmd(**lk)
        for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"):
            assert l[k] == o[k], (k, l[k], o[k])
        assert f32(l["lpmd"]).view(np.uint32) == f32(o["lpmd"]).view(np.uint32)


def test_density_chooser_takes_the_wide_form_on_sparse_batches(eng, kernel_form):
    """the per-batch choice between the dense tile kernel and the hashed-site form (launch_pdr_lpmd) is what production
    runs; with MTH_PDR_WIDE unset a WGBS-density batch must take k_pdr_lpmd_wide and a config-2-density batch the dense kernel
    (seen through the engine's per-kernel timers), both equal to the oracle"""
    if kernel_form != "auto":
        pytest.skip("the chooser only runs when MTH_PDR_WIDE is unset")
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(91)
    kw = dict(min_depth=3, min_cpgs=4, min_qual=10)        # (the CLI's min_cpgs: the chooser weighs the expected insertions per read)
    for dens, n_reads, want in ((0.0091, 200_000, "k_pdr_lpmd_wide"), (0.02, 500_000, "k_pdr_lpmd_tile")):
        c = synth.make_contig(0, 3_000_000, n_reads, dens, rng)
        reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
        eng.timing_enable(True)
        eng.timing_reset()
        p, l = run_device(eng, [c], PdrLpmdParams(**kw), device="cuda:0")
        t = eng.timing()
        eng.timing_enable(False)
        eng.timing_reset()
        ran = {k for k, v in t.items() if v[1] > 0 and k.startswith("k_pdr_lpmd")}
        assert ran == {want}, (dens, t)
        check_against_oracle(p, l, reads, kw, dict())


def test_read_clusters_with_megabase_gaps(eng):
    """reads in clusters with megabases of nothing between them (assembly gaps, capture panels): the fine read index is built by the
    one-loop fast form where a thread's four reads open few entries and by the general form where a wave meets a gap (round 5), the
    tile-granular families skip empty tiles; PDR + LPMD in every kernel form, MHL, ME / PM and FDRP / qFDRP against the oracle"""
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(77)
    length = 10_000_000
    starts = np.sort(np.concatenate([
        rng.integers(0, 60_000, size=6_000),                    # a dense cluster at the contig's start
        rng.integers(5_200_000, 6_000_000, size=3_000),         # a sparse stretch 5 Mbp further
        rng.integers(6_000_100, 6_000_400, size=300),           # a pile right behind it
        np.array([9_900_000]),                                  # one read near the end
    ])).astype(np.int32)
    c = synth.make_contig(1, length, len(starts), 0.03, rng, starts=starts)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    for kw in (dict(min_depth=3, min_cpgs=2, min_qual=10), dict(min_depth=0, min_cpgs=0, min_qual=0)):
        p, l = run_device(eng, [c], PdrLpmdParams(**kw), device="cuda:0")
        check_against_oracle(p, l, reads, kw, dict())
        assert len(p["pos"]) > 500
    bt = util.device_batch(c, device="cuda:0")
    eng.reset(); eng.mhl_accumulate(bt, min_depth=3, min_cpgs=2)
    d, o = eng.mhl_fetch(), reads.mhl(min_depth=3, min_cpgs=2)
    assert (d["pos"] == o.pos[:, 0]).all() and np.abs(d["mhl"].astype(np.float64) - o.val).max() <= 1e-6 and len(o) > 100
    eng.reset(); eng.quartet_accumulate(bt)
    d, o = eng.quartet_fetch(min_depth=2), reads.pm(min_depth=2)
    assert len(d["pm"]) == len(o) and (np.sort(d["pm"].view(np.uint32)) == np.sort(o.val.view(np.uint32))).all()
    eng.reset(); eng.fdrp_accumulate(bt, min_depth=2)
    d, o, q = eng.fdrp_fetch(), reads.fdrp(min_depth=2), reads.qfdrp(min_depth=2)
    assert (d["pos"] == o.pos[:, 0]).all() and (d["fdrp"].view(np.uint32) == o.val.view(np.uint32)).all() and (d["qfdrp"].view(np.uint32) == q.val.view(np.uint32)).all()
