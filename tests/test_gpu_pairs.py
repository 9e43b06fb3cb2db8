"""GPU parity of the LPMD per-pair table (`lpmd --pairs`; lpmd.rs:70-122) against the CPU oracle:
keys, counts and the per-pair f32 lpmd (lpmd.rs:111) all bit-exact, rows in the reference's sorted order."""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


def run_device(eng, contigs, kw, regions=None, device=None):
    from metheor_amd import shard
    eng.reset()
    keep = []
    for ci, c in enumerate(contigs):
        regs = regions[ci] if regions else [(0, c["length"])]
        for (b, e) in regs:
            sub = shard.slice_region(c, b, e) if regions else c
            bt = util.device_batch(sub, region=(b, e), device=device)
            keep.append(bt)
            eng.lpmd_pairs_accumulate(bt, **kw)
    return eng.lpmd_pairs_fetch()


def check(d, reads, kw):
    t = reads.lpmd(pairs=True, **kw)["pairs"]
    assert len(d["tid"]) == len(t)
    assert (d["tid"] == t.tid).all() and (d["pos1"] == t.pos[:, 0]).all() and (d["pos2"] == t.pos[:, 1]).all()
    assert (d["n_concordant"] == t.cnt[:, 0]).all() and (d["n_discordant"] == t.cnt[:, 1]).all()
    assert (d["lpmd"].view(np.uint32) == t.val.view(np.uint32)).all()
    return len(t)


def test_reference_fixtures(eng, golden_dir):
    for k in (1, 2, 3, 4, 5):
        rec = bamio.read_bam(os.path.join(golden_dir, "test%d.bam" % k))
        reads = pyoracle.Reads.decode(rec)
        c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
        kw = dict(min_distance=2, max_distance=16, min_qual=10)
        n = check(run_device(eng, [c], kw), reads, kw)
        # test1: z.z.z.z. -> pairs (0,2)(0,4)(0,6)(2,4)(2,6)(4,6); the summed counters give lpmd.rs:219's 48/48
        if k == 1:
            d = eng.lpmd_pairs_fetch()
            assert n == 6 and int(d["n_concordant"].sum()) == 48 and int(d["n_discordant"].sum()) == 48
        if k == 5:
            assert n == 0


def test_rrbs_and_parameters(eng, golden_dir):
    rec = bamio.read_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    reads = pyoracle.Reads.decode(rec)
    c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
    for kw in (dict(min_distance=2, max_distance=16, min_qual=10), dict(min_distance=1, max_distance=3, min_qual=10),
               dict(min_distance=5, max_distance=4, min_qual=10), dict(min_distance=0, max_distance=200, min_qual=43)):
        check(run_device(eng, [c], kw), reads, kw)
    assert check(run_device(eng, [c], dict(min_distance=2, max_distance=16, min_qual=10)), reads, dict(min_distance=2, max_distance=16, min_qual=10)) > 50


def test_synthetic_multi_contig_and_region_split(eng):
    from metheor_amd import shard, synth
    rng = np.random.default_rng(71)
    cs = [synth.make_contig(0, 300_000, 60_000, 0.03, rng), synth.make_contig(1, 500_000, 120_000, 0.03, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    kw = dict(min_distance=2, max_distance=16, min_qual=10)
    n = check(run_device(eng, cs, kw), reads, kw)
    assert n > 10000
    # a pair is owned by the region holding cpg1: region-split batches give the same rows, none twice
    check(run_device(eng, cs, kw, regions=[shard.plan_regions(cs[0], 3), shard.plan_regions(cs[1], 5)]), reads, kw)
    check(run_device(eng, cs, kw, device="cuda:0"), reads, kw)
    # the table is consistent with the global counters of the fused pass
    from metheor_amd import PdrLpmdParams
    d = run_device(eng, cs, kw)
    eng.reset()
    for c in cs:
        eng.pdr_lpmd_accumulate(util.device_batch(c), PdrLpmdParams(want_pdr=False))
    g = eng.lpmd_global()
    assert int(d["n_concordant"].sum()) == g["n_concordant"] and int(d["n_discordant"].sum()) == g["n_discordant"]


def test_table_overflow_retry(eng, monkeypatch):
    """the pairs table starts smaller than the number of pair updates; running out of probes redoes the pass 4x larger
    (forced by starting from 16 slots)"""
    from metheor_amd import synth
    c = synth.make_contig(0, 200_000, 30_000, 0.05, np.random.default_rng(78))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_distance=2, max_distance=40, min_qual=10)
    monkeypatch.setenv("MTH_PAIRS_SLOTS_MIN", "16")
    monkeypatch.setenv("MTH_PAIRS_FORCE_GLOBAL", "1")          # the global table only serves tiles the LDS table refused
    d = run_device(eng, [c], kw)
    monkeypatch.delenv("MTH_PAIRS_SLOTS_MIN")
    monkeypatch.delenv("MTH_PAIRS_FORCE_GLOBAL")
    assert check(d, reads, kw) > 2000


def test_output_redo_when_rows_do_not_fit(eng, monkeypatch):
    """the row buffer is sized from a guess; a batch that needs more is redone once with the exact size the kernel reports"""
    from metheor_amd import shard, synth
    rng = np.random.default_rng(79)
    cs = [synth.make_contig(0, 300_000, 50_000, 0.04, rng), synth.make_contig(1, 200_000, 30_000, 0.04, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    kw = dict(min_distance=2, max_distance=30, min_qual=10)
    monkeypatch.setenv("MTH_PAIRS_ROWS_MIN", "100")
    d = run_device(eng, cs, kw, regions=[shard.plan_regions(cs[0], 3), [(0, cs[1]["length"])]])
    monkeypatch.delenv("MTH_PAIRS_ROWS_MIN")
    assert check(d, reads, kw) > 10000


def test_global_path_for_every_tile(eng, monkeypatch):
    from metheor_amd import shard, synth
    rng = np.random.default_rng(80)
    cs = [synth.make_contig(0, 300_000, 50_000, 0.04, rng), synth.make_contig(1, 200_000, 30_000, 0.04, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    kw = dict(min_distance=2, max_distance=16, min_qual=10)
    monkeypatch.setenv("MTH_PAIRS_FORCE_GLOBAL", "1")
    d = run_device(eng, cs, kw, regions=[shard.plan_regions(cs[0], 3), [(0, cs[1]["length"])]])
    monkeypatch.delenv("MTH_PAIRS_FORCE_GLOBAL")
    assert check(d, reads, kw) > 5000


def test_dense_and_wide_window_overflow_the_lds_table(eng):
    """density 0.2 with a 100-bp window: > 2048 distinct pairs per 8192-bp tile in the dense contig -> those tiles take the
    global path, the sparse contig stays on the tile path; rows still come out in the reference's sorted order"""
    from metheor_amd import synth
    rng = np.random.default_rng(81)
    cs = [synth.make_contig(0, 100_000, 20_000, 0.2, rng), synth.make_contig(1, 200_000, 30_000, 0.01, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    kw = dict(min_distance=2, max_distance=100, min_qual=10)
    assert check(run_device(eng, cs, kw), reads, kw) > 50000


def test_deep_tile_exceeds_16bit_counters(eng):
    from metheor_amd import synth
    from tests.test_gpu_quartet import _contig, _reads_over
    rng = np.random.default_rng(82)
    sites = np.array([5000, 5004, 5010, 5030, 5040, 9000, 9004, 9010, 9100, 9150], np.int64)
    starts = np.sort(np.r_[np.full(70_000, 4990), rng.integers(8950, 9000, 3000)]).astype(np.int64)
    off, pos, rel = _reads_over(sites, starts, 120, rng)
    c = _contig(0, 20000, starts, 120, off, pos, rel, rng)
    c["read_mapq"][:] = 40
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_distance=2, max_distance=16, min_qual=10)
    d = run_device(eng, [c], kw)
    check(d, reads, kw)
    assert int((d["n_concordant"] + d["n_discordant"]).max()) == 70_000


@pytest.mark.parametrize("shift", [13, 14, 15])
def test_every_tile_width(eng, monkeypatch, shift):
    """the tile width is chosen per batch from its call density; each of the three widths gives the same rows"""
    from metheor_amd import shard, synth
    rng = np.random.default_rng(60 + shift)
    cs = [synth.make_contig(0, 400_000, 60_000, 0.012, rng), synth.make_contig(1, 150_000, 20_000, 0.05, rng)]
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    kw = dict(min_distance=2, max_distance=30, min_qual=10)
    monkeypatch.setenv("MTH_PAIRS_TILE_SHIFT", str(shift))
    d = run_device(eng, cs, kw, regions=[shard.plan_regions(cs[0], 2), [(0, cs[1]["length"])]])
    monkeypatch.delenv("MTH_PAIRS_TILE_SHIFT")
    assert check(d, reads, kw) > 8000


# ---- batches queued without a host sync (device-resident batches after a context's first; pairs_batch / pairs_resolve) ----
@pytest.mark.parametrize("force", ["", "global", "unfit"])
def test_queued_batches(monkeypatch, capfd, force):
    """global: every tile of the queued batches refuses its LDS table -> replayed through the global path; unfit: the context learns
    its output sizing from a tiny batch, the real ones behind it do not fit what they are given -> replayed at their exact size"""
    import re
    import metheor_amd
    from metheor_amd import synth
    rng = np.random.default_rng(41)
    if force == "unfit":
        cs = [synth.make_contig(0, 4_000, 40, 0.002, rng)] + [synth.make_contig(t, 1_200_000, 25_000, 0.04, rng) for t in (1, 2, 3)]
    else:
        cs = [synth.make_contig(t, 50_000 + 7_000 * t, 5_000 + 900 * t, 0.04, rng) for t in range(5)]
    kw = dict(min_distance=2, max_distance=16, min_qual=10)
    tabs = [pyoracle.Reads.decode(util.contig_to_records(c, "ctg%d" % c["tid"])).lpmd(pairs=True, **kw)["pairs"] for c in cs]
    monkeypatch.setenv("MTH_PAIRS_DEBUG", "1")
    e = metheor_amd.Engine(0)
    try:
        bts = [util.device_batch(c, device="cuda:0") for c in cs]
        e.lpmd_pairs_accumulate(bts[0], **kw)                  # synchronous: a fresh context
        if force == "global":
            monkeypatch.setenv("MTH_PAIRS_FORCE_GLOBAL", "1")
        for bt in bts[1:3]:
            e.lpmd_pairs_accumulate(bt, **kw)
        capfd.readouterr()
        e.sync()                                                # resolves; more batches after it
        q, r = map(int, re.search(r"\[pairs\] queued batches (\d+), replayed (\d+)", capfd.readouterr().err).groups())
        assert q == 2 and r == (0 if force == "" else 2), (q, r)
        for bt in bts[3:]:
            e.lpmd_pairs_accumulate(bt, **kw)
        d = e.lpmd_pairs_fetch()
        off = 0
        for c, t in zip(cs, tabs):
            m = slice(off, off + len(t))
            assert (d["tid"][m] == c["tid"]).all() and (d["pos1"][m] == t.pos[:, 0]).all() and (d["pos2"][m] == t.pos[:, 1]).all()
            assert (d["n_concordant"][m] == t.cnt[:, 0]).all() and (d["n_discordant"][m] == t.cnt[:, 1]).all()
            assert (d["lpmd"][m].view(np.uint32) == t.val.view(np.uint32)).all()
            off += len(t)
        assert off == len(d["tid"])
    finally:
        e.close()


def test_queued_batch_beyond_its_estimate_is_replayed_not_truncated(monkeypatch, capfd):
    """ADVICE r04 (high), the pairs table's half: a queued batch that produces more rows than its estimate but still fits the buffer
    (grown to 1.25 x the estimate) used to lose the rows beyond the estimate when the next queued batch grew the buffer.  It is unfit
    now and the resolve replays it (mth_pairs.hip, pairs_batch)."""
    import metheor_amd
    from metheor_amd import synth
    rng = np.random.default_rng(2025)
    cs = [synth.make_contig(t, ln, nr, 0.04, rng) for t, (ln, nr) in enumerate([(100_000, 10_000), (600_000, 60_000), (900_000, 90_000)])]
    kw = dict(min_distance=2, max_distance=16, min_qual=10)
    tabs = [pyoracle.Reads.decode(util.contig_to_records(c, "ctg%d" % c["tid"])).lpmd(pairs=True, **kw)["pairs"] for c in cs]
    monkeypatch.setenv("MTH_PAIRS_DEBUG", "1")
    e = metheor_amd.Engine(0)
    try:
        bts = [util.device_batch(c, device="cuda:0") for c in cs]
        e.lpmd_pairs_accumulate(bts[0], **kw)                                   # synchronous: a fresh context
        monkeypatch.setenv("MTH_PAIRS_ROWS_MIN", str(int(len(tabs[1]) / 1.1)))  # B: estimate < rows <= 1.25 x estimate
        e.lpmd_pairs_accumulate(bts[1], **kw)
        monkeypatch.setenv("MTH_PAIRS_ROWS_MIN", str(2 * len(tabs[2]) + 100_000))   # C: forces the buffer to grow
        e.lpmd_pairs_accumulate(bts[2], **kw)
        monkeypatch.delenv("MTH_PAIRS_ROWS_MIN")
        capfd.readouterr()
        d = e.lpmd_pairs_fetch()
        err = capfd.readouterr().err
        assert "[pairs] queued batches 2, replayed" in err and "replayed 0" not in err, err
        off = 0
        for c, t in zip(cs, tabs):
            m = slice(off, off + len(t))
            assert (d["tid"][m] == c["tid"]).all() and (d["pos1"][m] == t.pos[:, 0]).all() and (d["pos2"][m] == t.pos[:, 1]).all()
            assert (d["n_concordant"][m] == t.cnt[:, 0]).all() and (d["n_discordant"][m] == t.cnt[:, 1]).all()
            off += len(t)
        assert off == len(d["tid"])
    finally:
        e.close()
