"""Prepared batches (include/metheor_hip.h, mth_batch_prepare / mth_batch_release): one device-resident copy and ONE read index per batch,
shared by every measure's entry point.  Each measure replaces one compute_helper pass of the reference over the same file (pdr.rs:119,
lpmd.rs:154, me.rs:90, pm.rs:85, mhl.rs:135, fdrp.rs:176, qfdrp.rs:188): the rows from a prepared batch must be the rows of the plain
entry points -- which the per-measure suites pin against the oracle -- bit for bit, and the oracle's own where it is cheap to ask."""
import numpy as np
import pytest

from oracle import pyoracle
from tests import util

pytestmark = pytest.mark.gpu


def _all_seven(eng, batches, seed=3):
    from metheor_amd import PdrLpmdParams
    out = {}
    eng.reset()
    for b in batches:
        eng.pdr_lpmd_accumulate(b, PdrLpmdParams(min_depth=3, min_cpgs=2))
    out["pdr"] = eng.pdr_fetch(); out["lpmd"] = eng.lpmd_global()
    for b in batches:
        eng.quartet_accumulate(b, min_qual=10)
    out["quartet"] = eng.quartet_fetch(min_depth=2)
    for b in batches:
        eng.mhl_accumulate(b, min_depth=3, min_cpgs=2)
    out["mhl"] = eng.mhl_fetch()
    for b in batches:
        eng.fdrp_accumulate(b, min_depth=3, seed=seed)
    out["fdrp"] = eng.fdrp_fetch()
    for b in batches:
        eng.lpmd_pairs_accumulate(b)
    out["pairs"] = eng.lpmd_pairs_fetch()
    return out


def _same(a, b):
    for m in a:
        for k in a[m]:
            x, y = np.asarray(a[m][k]), np.asarray(b[m][k])
            assert x.shape == y.shape, (m, k, x.shape, y.shape)
            if x.dtype.kind == "f":
                assert (x.view(np.uint32) == y.view(np.uint32)).all() or (np.isnan(x) == np.isnan(y)).all() and np.array_equal(x[~np.isnan(x)], y[~np.isnan(y)]), (m, k)
            else:
                assert (x == y).all(), (m, k)


@pytest.mark.parametrize("device_mem", [True, False])
@pytest.mark.parametrize("runs", [False, True])
def test_all_seven_measures_from_one_prepared_batch(monkeypatch, device_mem, runs):
    """two contigs (one dense, one at WGBS density so that the wide PDR form and walk4 are taken), each prepared ONCE, then all seven
    measures + the pairs table from the prepared batches: identical to the plain entry points; PDR / LPMD also against the oracle"""
    import metheor_amd
    from metheor_amd import synth
    if runs:
        monkeypatch.setenv("MTH_TILE_RUNS", "1")
    else:
        monkeypatch.delenv("MTH_TILE_RUNS", raising=False)
    rng = np.random.default_rng(17)
    cs = [synth.make_contig(0, 600_000, 90_000, 0.03, rng), synth.make_contig(1, 3_000_000, 200_000, 0.0091, rng)]
    dev = "cuda:0" if device_mem else None
    eng = metheor_amd.Engine(0)
    try:
        plain = [util.device_batch(c, device=dev) for c in cs]
        want = _all_seven(eng, plain)
        prepared = [eng.batch_prepare(b) for b in plain]
        got = _all_seven(eng, prepared)
        _same(want, got)
        got2 = _all_seven(eng, prepared)                  # and again: nothing of the first job is left in the handles
        _same(want, got2)
        reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
        o = reads.pdr(min_depth=3, min_cpgs=2, min_qual=10)
        assert (got["pdr"]["pos"] == o.pos[:, 0]).all() and (got["pdr"]["n_concordant"] == o.cnt[:, 0]).all() and (got["pdr"]["n_discordant"] == o.cnt[:, 1]).all()
        ol = reads.lpmd()
        assert all(got["lpmd"][k] == ol[k] for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"))
        for p in prepared:
            p.release()
        # a released handle is refused, the original batch still works
        with pytest.raises(metheor_amd.MthError):
            eng.pdr_lpmd_accumulate(prepared[0], metheor_amd.PdrLpmdParams())
        eng.reset()
        eng.pdr_lpmd_accumulate(plain[0], metheor_amd.PdrLpmdParams(min_depth=3, min_cpgs=2))
        assert eng.pdr_count() > 0
    finally:
        eng.close()


def test_prepared_unsorted_batch_is_reported_by_every_measure():
    """the sortedness check runs once, at prepare time; every measure that uses the prepared batch still reports it (also after a reset)"""
    import metheor_amd
    from metheor_amd import synth
    rng = np.random.default_rng(5)
    c = synth.make_contig(0, 200_000, 20_000, 0.03, rng)
    c = dict(c)
    rs = c["read_start"].copy(); rs[1000], rs[1001] = rs[1001] + 500, rs[1000]        # out of order
    c["read_start"] = rs
    eng = metheor_amd.Engine(0)
    try:
        b = util.device_batch(c, device="cuda:0")
        p = eng.batch_prepare(b)
        for k in range(2):
            eng.reset()
            eng.pdr_lpmd_accumulate(p, metheor_amd.PdrLpmdParams())
            with pytest.raises(metheor_amd.MthError) as ei:
                eng.pdr_count()
            assert "sorted" in str(ei.value)
        eng.reset()
        eng.quartet_accumulate(p)
        with pytest.raises(metheor_amd.MthError):
            eng.quartet_fetch()
        eng.reset()
        p.release()
    finally:
        eng.close()


def test_prepared_batch_of_another_context_is_refused():
    import metheor_amd
    from metheor_amd import synth
    rng = np.random.default_rng(6)
    c = synth.make_contig(0, 100_000, 5_000, 0.03, rng)
    e1, e2 = metheor_amd.Engine(0), metheor_amd.Engine(0)
    try:
        p = e1.batch_prepare(util.device_batch(c, device="cuda:0"))
        with pytest.raises(metheor_amd.MthError):
            e2.pdr_lpmd_accumulate(p, metheor_amd.PdrLpmdParams())
        p.release()
    finally:
        e1.close(); e2.close()


def test_prepared_batches_are_queued_and_pipelined_like_device_batches(monkeypatch, capfd):
    """ME / PM and pairs batches after a context's first are queued without a host sync, consecutive PDR + LPMD batches are pipelined on two
    lanes -- prepared batches take both paths; a replayed queued batch rebuilds its own index (the handle current at replay time is
    another batch's).  Six contigs, every batch prepared; rows equal to the plain device batches'."""
    import metheor_amd
    from metheor_amd import PdrLpmdParams, synth
    rng = np.random.default_rng(23)
    cs = [synth.make_contig(t, 200_000 + 37_000 * t, 30_000 + 4_000 * t, 0.03, rng) for t in range(6)]
    monkeypatch.setenv("MTH_QUARTET_DEBUG", "1")
    monkeypatch.setenv("MTH_PAIRS_DEBUG", "1")
    out = {}
    for prepared in (False, True):
        eng = metheor_amd.Engine(0)
        try:
            plain = [util.device_batch(c, device="cuda:0") for c in cs]
            bts = [eng.batch_prepare(b) for b in plain] if prepared else plain
            eng.reset()
            for b in bts:
                eng.pdr_lpmd_accumulate(b, PdrLpmdParams(min_depth=3, min_cpgs=2))
            r = {"pdr": eng.pdr_fetch(), "lpmd": eng.lpmd_global()}
            for b in bts:
                eng.quartet_accumulate(b, min_qual=10)
            capfd.readouterr()
            r["quartet"] = eng.quartet_fetch(min_depth=2)
            err = capfd.readouterr().err
            assert "[quartet] queued batches" in err and "queued batches 0" not in err, err
            # force a replay: every tile of the third batch refuses its LDS table (its snapshot says so, the resolve replays it)
            eng.reset()
            for k, b in enumerate(bts):
                if k == 2:
                    monkeypatch.setenv("MTH_PAIRS_FORCE_GLOBAL", "1")
                eng.lpmd_pairs_accumulate(b)
                monkeypatch.delenv("MTH_PAIRS_FORCE_GLOBAL", raising=False)
            capfd.readouterr()
            r["pairs"] = eng.lpmd_pairs_fetch()
            err = capfd.readouterr().err
            assert "[pairs] queued batches" in err and "replayed 0" not in err, err
            out[prepared] = r
            if prepared:
                for b in bts:
                    b.release()
        finally:
            eng.close()
    _same(out[False], out[True])


def test_prepared_group_and_region_batches():
    """the two other shapes a batch takes -- a contig GROUP (several contigs in one virtual coordinate space, mth_group_define) and
    REGION pieces of one contig (region_beg > 0, halo reads before the region) -- prepared once each: every measure equals the plain
    entry points bit for bit, and PDR / MHL the oracle's rows"""
    import metheor_amd
    from metheor_amd import shard, synth
    from tests import test_gpu_groups as G
    cs = G._contigs(21, n=4)
    voff, v = [], 0
    for c in cs:
        voff.append(v); v += ((c["length"] + 150 + 1024 + 4095) // 4096) * 4096
    tids = [2, 4, 9, 10]
    for c, t in zip(cs, tids):
        c["tid"] = t
    rng = np.random.default_rng(5)
    big = synth.make_contig(0, 500_000, 60_000, 0.03, rng)
    cuts = [(0, 170_000), (170_000, 333_333), (333_333, 500_000)]
    eng = metheor_amd.Engine(0)
    try:
        h = eng.group_define(tids, voff)
        plain = [G._group_batch(cs, voff, h, "cuda:0")]
        want = _all_seven(eng, plain)
        prepared = [eng.batch_prepare(b) for b in plain]
        _same(want, _all_seven(eng, prepared))
        ora = G._oracle(cs)
        t_, p_, v_, c_ = G._cat([o.mhl(min_depth=3, min_cpgs=2) for o in ora], tids)
        assert (want["mhl"]["tid"] == t_).all() and (want["mhl"]["pos"] == p_[:, 0]).all()
        for p in prepared:
            p.release()
        # region pieces of one contig
        eng.reset()
        plain = [util.device_batch(shard.slice_region(big, b, e), region=(b, e), device="cuda:0") for (b, e) in cuts]
        want = _all_seven(eng, plain)
        prepared = [eng.batch_prepare(b) for b in plain]
        _same(want, _all_seven(eng, prepared))
        reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(big))
        o = reads.pdr(min_depth=3, min_cpgs=2, min_qual=10)
        assert (want["pdr"]["pos"] == o.pos[:, 0]).all() and (want["pdr"]["n_concordant"] == o.cnt[:, 0]).all()
        o = reads.mhl(min_depth=3, min_cpgs=2)
        assert (want["mhl"]["pos"] == o.pos[:, 0]).all() and np.abs(want["mhl"]["mhl"].astype(np.float64) - o.val).max() <= 1e-6
    finally:
        eng.close()


def test_handle_validation_by_registry_and_free_on_close():
    """a prepared batch's handle is validated by look-up in the context's registry, never by dereferencing it -- a copy
    used after release, a second release and an edited copy are refused; what the caller does not release, the context's close frees
    (a release after the close is a no-op, not a crash)"""
    import copy
    import metheor_amd
    from metheor_amd import synth
    c = synth.make_contig(0, 500_000, 60_000, 0.02, np.random.default_rng(5))
    eng = metheor_amd.Engine(0)
    bt = util.device_batch(c, device="cuda:0")
    p = eng.batch_prepare(bt)
    eng.pdr_lpmd_accumulate(p, metheor_amd.PdrLpmdParams())
    n0 = len(eng.pdr_fetch()["pos"])
    assert n0 > 100
    stale = copy.copy(p)                                   # the caller's own copy of the struct
    stale.c = type(p.c).from_buffer_copy(p.c)
    # an edited copy: the entry points size their outputs from the caller's struct
    edited = copy.copy(p)
    edited.c = type(p.c).from_buffer_copy(p.c)
    edited.c.n_cpgs = p.c.n_cpgs // 2
    with pytest.raises(metheor_amd.MthError) as ei:
        eng.pdr_lpmd_accumulate(edited, metheor_amd.PdrLpmdParams())
    assert ei.value.status == -1
    p.release()
    with pytest.raises(metheor_amd.MthError):              # the copy still carries the pointer: refused by look-up
        eng.pdr_lpmd_accumulate(stale, metheor_amd.PdrLpmdParams())
    with pytest.raises(metheor_amd.MthError):
        stale.eng._check(stale.eng.L.mth_batch_release(stale.eng.h, metheor_amd.capi.C.byref(stale.c)))
    # outstanding prepared batches at close: freed by the context
    q1, q2 = eng.batch_prepare(bt), eng.batch_prepare(bt)
    eng.mhl_accumulate(q1)
    assert len(eng.mhl_fetch()["pos"]) > 0
    eng.close()
    q1.release(); q2.release()                             # after the close: nothing to do
