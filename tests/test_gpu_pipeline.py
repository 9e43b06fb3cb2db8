"""GPU parity of the PDR + LPMD batch pipeline (metheor_amd/csrc/mth_pdr_lpmd.hip, "Pipelined batches"): consecutive device-resident
batches alternate between two lanes (a stream and a set of work buffers each), only their gathers form a chain, and a mth_reset
inside such a run is folded into the next batch's gather.  What must not change: rows, counters, error reporting -- against the
oracle (pdr.rs:119-212, lpmd.rs:154-202) and against the same calls with the pipeline switched off (MTH_PIPELINE=0)."""
import numpy as np
import pytest

from oracle import pyoracle
from tests import util

pytestmark = pytest.mark.gpu


def _engine(monkeypatch, pipeline):
    import metheor_amd
    monkeypatch.setenv("MTH_PIPELINE", "1" if pipeline else "0")      # read at the context's first PDR + LPMD call
    monkeypatch.delenv("MTH_PDR_WIDE", raising=False)
    return metheor_amd.Engine(0)


def _contigs(seed, n=7):
    from metheor_amd import synth
    rng = np.random.default_rng(seed)
    shapes = [(300_000, 40_000, 0.03), (1_000_000, 150_000, 0.02), (50_000, 3_000, 0.05), (2_000_000, 120_000, 0.009),
              (20_000, 30_000, 0.04), (700_000, 90_000, 0.02), (4_100, 900, 0.03), (5_000_000, 400_000, 0.0091)]
    return [synth.make_contig(t, ln, nr, dens, rng) for t, (ln, nr, dens) in enumerate(shapes[:n])]


def _same(a, b):
    return all((a[k].view(np.uint32) == b[k].view(np.uint32)).all() for k in a)


def _submit(eng, cs, p, keep, regions=None):
    from metheor_amd import shard
    for ci, c in enumerate(cs):
        regs = regions[ci] if regions else [(0, c["length"])]
        for (b, e) in regs:
            sub = shard.slice_region(c, b, e) if regions else c
            bt = util.device_batch(sub, region=(b, e), device="cuda:0")
            keep.append(bt)
            eng.pdr_lpmd_accumulate(bt, p)


@pytest.mark.parametrize("kw", [dict(min_depth=10, min_cpgs=4, min_qual=10), dict(min_depth=0, min_cpgs=0, min_qual=0),
                                dict(min_depth=3, min_cpgs=1, min_qual=10, want_lpmd=False), dict(want_pdr=False)])
@pytest.mark.parametrize("runs", [False, True])
def test_pipelined_job_equals_serial_and_oracle(monkeypatch, kw, runs):
    """runs: the dense batches through the persistent run form (k_pdr_lpmd_runs + k_gather_runs, MTH_TILE_RUNS=1) -- its gather is a
    link of the same chain"""
    from metheor_amd import PdrLpmdParams, shard, synth
    if runs:
        monkeypatch.setenv("MTH_TILE_RUNS", "1")
    else:
        monkeypatch.delenv("MTH_TILE_RUNS", raising=False)
    cs = _contigs(101)
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    regions = [[(0, c["length"])] for c in cs]
    regions[1] = shard.plan_regions(cs[1], 5)
    regions[6] = shard.plan_regions(cs[6], 2)
    p = PdrLpmdParams(**kw)
    out = {}
    for pipeline in (True, False):
        eng = _engine(monkeypatch, pipeline)
        keep = []
        eng.reset()
        _submit(eng, cs, p, keep, regions)
        out[pipeline] = (eng.pdr_fetch(), eng.lpmd_global())
        eng.close()
    (d1, l1), (d0, l0) = out[True], out[False]
    assert _same(d1, d0) and all(l1[k] == l0[k] for k in l1 if k != "lpmd")
    if kw.get("want_pdr", True):
        o = reads.pdr(**{k: v for k, v in kw.items() if k.startswith("min_")})
        assert len(o) == len(d1["pos"]) and (d1["pos"] == o.pos[:, 0]).all() and (d1["tid"] == o.tid).all()
        assert (d1["n_concordant"] == o.cnt[:, 0]).all() and (d1["n_discordant"] == o.cnt[:, 1]).all()
        assert (d1["pdr"].view(np.uint32) == o.val.view(np.uint32)).all()
    else:
        assert len(d1["pos"]) == 0
    if kw.get("want_lpmd", True):
        ol = reads.lpmd()
        assert all(l1[k] == ol[k] for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"))


def test_reset_between_jobs_without_a_sync(monkeypatch):
    """job after job on one context, mth_reset in between and no synchronising call until the end: the last job's result is the
    job's own (nothing of the earlier ones leaks through the folded reset), and a getter in the middle sees the middle"""
    from metheor_amd import PdrLpmdParams
    cs = _contigs(202, n=5)
    p = PdrLpmdParams(min_depth=5, min_cpgs=2)
    eng0 = _engine(monkeypatch, False)
    ref = []
    for job in range(4):
        keep = []
        eng0.reset()
        _submit(eng0, cs[job % 3:job % 3 + 2], p, keep)
        ref.append((eng0.pdr_fetch(), eng0.lpmd_global()))
    eng0.close()
    eng = _engine(monkeypatch, True)
    keep = []
    for job in range(4):
        eng.reset()
        _submit(eng, cs[job % 3:job % 3 + 2], p, keep)
        if job == 1:
            d, l = eng.pdr_fetch(), eng.lpmd_global()
            assert _same(d, ref[1][0]) and all(l[k] == ref[1][1][k] for k in l if k != "lpmd")
    d, l = eng.pdr_fetch(), eng.lpmd_global()
    assert _same(d, ref[3][0]) and all(l[k] == ref[3][1][k] for k in l if k != "lpmd")
    # a reset that nothing follows: an empty job
    eng.reset()
    assert eng.pdr_count() == 0 and eng.lpmd_global()["n_read"] == 0
    # the bench's loop: reset + one batch, many times, one sync
    for _ in range(25):
        eng.reset()
        _submit(eng, cs[1:2], p, keep)
    d1 = eng.pdr_fetch()
    eng.reset()
    _submit(eng, cs[1:2], p, keep)
    assert _same(d1, eng.pdr_fetch())
    eng.close()


def test_errors_surface_and_do_not_outlive_a_reset(monkeypatch):
    """an unsorted batch in the middle of a pipelined run fails the next getter (MTH_ERR_UNSORTED, as without the pipeline); after
    mth_reset the context is clean, also when the reset was folded into a gather"""
    from metheor_amd import MthError, PdrLpmdParams
    cs = _contigs(303, n=4)
    bad = dict(cs[2])
    rs = bad["read_start"].copy()
    rs[100], rs[2000] = rs[2000], rs[100]
    bad["read_start"] = rs
    p = PdrLpmdParams(min_depth=0, min_cpgs=0)
    eng = _engine(monkeypatch, True)
    keep = []
    eng.reset()
    _submit(eng, [cs[0], cs[1], bad, cs[3]], p, keep)
    with pytest.raises(MthError):
        eng.pdr_fetch()
    # reset, then straight into a clean pipelined job (the reset is folded into its first gather only from the second call of a
    # run on; both orders are exercised: after the failing getter the run restarts)
    eng.reset()
    _submit(eng, [cs[0], cs[1], cs[3]], p, keep)
    d = eng.pdr_fetch()
    eng.reset()
    _submit(eng, [cs[0], cs[1], bad], p, keep)      # error raised inside a run ...
    eng.reset()                                     # ... reset while the lanes are busy ...
    _submit(eng, [cs[0], cs[1], cs[3]], p, keep)    # ... and a clean job right behind it
    assert _same(d, eng.pdr_fetch())
    eng.close()


def test_other_measures_and_host_batches_join_the_lanes(monkeypatch):
    """any other entry point orders ctx->stream behind the lanes: PDR batches, then ME / PM on the same context, then PDR again,
    with host-resident batches mixed in (they always run unpipelined)"""
    from metheor_amd import PdrLpmdParams, synth
    cs = _contigs(404, n=4)
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    p = PdrLpmdParams(min_depth=2, min_cpgs=2)
    eng = _engine(monkeypatch, True)
    keep = []
    eng.reset()
    _submit(eng, cs[:3], p, keep)
    hb = util.device_batch(cs[3])                   # host numpy arrays
    eng.pdr_lpmd_accumulate(hb, p)
    for c in cs:
        bt = util.device_batch(c, device="cuda:0")
        keep.append(bt)
        eng.quartet_accumulate(bt, min_qual=10)
    d, l = eng.pdr_fetch(), eng.lpmd_global()
    o, ol = reads.pdr(min_depth=2, min_cpgs=2), reads.lpmd()
    assert len(o) == len(d["pos"]) and (d["pos"] == o.pos[:, 0]).all() and (d["n_discordant"] == o.cnt[:, 1]).all()
    assert all(l[k] == ol[k] for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"))
    q = eng.quartet_fetch(min_depth=0)
    assert len(q["pos"]) == len(reads.me(min_depth=0, min_qual=10))
    eng.close()
