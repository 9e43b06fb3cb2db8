"""GPU parity, randomised: many small seeded scenarios, every measure against the CPU oracle through the C ABI.

The hand-built and fixture tests pin known semantics; this one looks for what nobody thought of.  Generators lean
on the edges the kernels special-case: reads without CpGs, duplicate starts, low mapq, both strands (a reverse
read's first call at start-1), calls dropped at random (a read that covers a site without calling it: flushes
and re-opened sites), dense CpGs (more calls than registers), long reads (spans > 150 / > 200 bp), shallow and
deep piles (reservoir sampling), random parameters, whole contigs and region slices.  Same bars as the
per-measure tests (their check functions are reused): integers bit-exact, floats bit-exact or within 1e-6."""
import numpy as np
import pytest

from oracle import pyoracle
from tests import test_gpu_fdrp as T_fdrp
from tests import test_gpu_mhl as T_mhl
from tests import test_gpu_pairs as T_pairs
from tests import test_gpu_pdr_lpmd as T_pdr
from tests import test_gpu_quartet as T_quartet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


def drop_calls(c, rng, frac):
    """remove a random subset of the CpG calls (mismatch / non-CpG context in the XM string)"""
    if frac <= 0 or len(c["cpg_pos"]) == 0:
        return c
    keep = rng.random(len(c["cpg_pos"])) >= frac
    cnt = np.diff(c["cpg_off"].astype(np.int64))
    read_of = np.repeat(np.arange(len(cnt)), cnt)
    new_cnt = np.bincount(read_of[keep], minlength=len(cnt))
    off = np.zeros(len(cnt) + 1, np.int64)
    np.cumsum(new_cnt, out=off[1:])
    d = dict(c)
    d["cpg_off"] = off.astype(np.uint32)
    d["cpg_pos"] = c["cpg_pos"][keep]
    d["cpg_rel"] = c["cpg_rel"][keep]
    return d


def scenario(seed):
    from metheor_amd import synth
    rng = np.random.default_rng(10_000 + seed)
    n_contigs = int(rng.integers(1, 3))
    read_len = int(rng.choice([36, 75, 100, 150, 150, 150, 200, 250]))
    density = float(rng.choice([0.004, 0.02, 0.02, 0.05, 0.12, 0.3]))
    cs = []
    for tid in range(n_contigs):
        length = int(rng.integers(2_000, 40_000))
        n_reads = int(rng.integers(20, 4_000))
        starts = None
        mode = int(rng.integers(0, 4))
        hi = max(length - read_len, 1)
        if mode == 1:      # piles: a few start positions, many duplicates
            spots = rng.integers(0, hi, size=int(rng.integers(1, 12)))
            starts = np.sort(rng.choice(spots, size=n_reads) + rng.integers(0, 3, size=n_reads)).astype(np.int32)
        elif mode == 2:    # one deep window inside a shallow background
            w0 = int(rng.integers(0, hi))
            starts = np.sort(np.concatenate([rng.integers(0, hi, size=n_reads // 3),
                                             rng.integers(w0, min(w0 + 300, hi) + 1, size=n_reads - n_reads // 3)])).astype(np.int32)
        c = synth.make_contig(tid, length, n_reads, density, rng, read_len=read_len,
                              low_mapq_frac=float(rng.choice([0.0, 0.05, 0.4])), starts=starts)
        cs.append(drop_calls(c, rng, float(rng.choice([0.0, 0.0, 0.1, 0.4]))))
    regions = None
    if rng.random() < 0.4:
        from metheor_amd import shard
        regions = [shard.plan_regions(c, int(rng.integers(2, 5))) for c in cs]
    return rng, cs, regions, read_len


def fdrp_checked(eng, cs, fk, regions, reads):
    """FDRP / qFDRP against the oracle with the engine's own choice of kernel form, then once more with each form forced --
    the one-pass tile form (k_fdrp_wtile, also from 256-position stretches, and switched off), k_fdrp_walk4 (four sites per wave,
    hand-back to the general walk), the read x read form (k_fdrp_tile + k_fdrp_chain), the
    wave-per-site walk alone: the same rows bit for bit"""
    import os
    d0 = T_fdrp.run_device(eng, cs, fk, regions=regions)
    T_fdrp.check(d0, reads, fk)
    for env in (dict(METHEOR_FDRP_WTILE="1"), dict(METHEOR_FDRP_WTILE="1", METHEOR_FDRP_WTILE_SUB="1"), dict(METHEOR_FDRP_WTILE="0"),
                dict(METHEOR_FDRP_WALK4="16", METHEOR_FDRP_TILE="0"), dict(METHEOR_FDRP_TILE="1"), dict(METHEOR_FDRP_TILE="1", METHEOR_FDRP_TILE_WIDE_ROWS="1"), dict(METHEOR_FDRP_WALK4="0", METHEOR_FDRP_TILE="0")):
        os.environ.update(env)
        try:
            d1 = T_fdrp.run_device(eng, cs, fk, regions=regions)
        finally:
            for k in env:
                del os.environ[k]
        assert len(d0["pos"]) == len(d1["pos"]), env
        for k in ("tid", "pos", "n_reads"):
            assert (d0[k] == d1[k]).all(), (k, env)
        for k in ("fdrp", "qfdrp"):
            assert (d0[k].view(np.uint32) == d1[k].view(np.uint32)).all(), (k, env)


@pytest.mark.parametrize("seed", range(96))
def test_random_scenario_all_measures(eng, seed):
    from metheor_amd import PdrLpmdParams, synth
    rng, cs, regions, read_len = scenario(seed)
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    mq = int(rng.choice([0, 10, 10, 43]))

    # PDR + LPMD (fused tile kernel for spans <= 150, exact site walk for the PDR half beyond)
    pk = dict(min_depth=int(rng.choice([0, 1, 3, 10])), min_cpgs=int(rng.choice([0, 1, 2, 4, 9])), min_qual=mq)
    lk = dict(min_distance=int(rng.choice([0, 1, 2, 5])), max_distance=int(rng.choice([1, 4, 16, 60, 300])), min_qual=mq)
    p = PdrLpmdParams(min_distance=lk["min_distance"], max_distance=lk["max_distance"], lpmd_min_qual=mq, **pk)
    rel16 = bool(rng.integers(0, 2))
    d, l = T_pdr.run_device(eng, cs, p, regions=regions, rel16=rel16)
    T_pdr.check_against_oracle(d, l, reads, pk, lk)
    # the same through the hashed-site form for sparse batches (mth_pdr_wide.hip) whatever the batch's density, and through the dense
    # form forced: identical rows and counters
    import os
    # ... and through the persistent run form of the dense kernel (round 5: k_pdr_lpmd_runs + k_gather_runs, MTH_TILE_RUNS=1; it takes
    # the batches with 8-bit relative positions, the others fall back to the tile form)
    for envs in ({"MTH_PDR_WIDE": str(14 + seed % 3)}, {"MTH_PDR_WIDE": "0"}, {"MTH_PDR_WIDE": "0", "MTH_TILE_RUNS": "1"}):
        os.environ.update(envs)
        try:
            d2, l2 = T_pdr.run_device(eng, cs, p, regions=regions, rel16=rel16)
        finally:
            for k_ in envs:
                del os.environ[k_]
        assert all((d[k].view(np.uint32) == d2[k].view(np.uint32)).all() for k in d) and all(l[k] == l2[k] for k in l if k != "lpmd"), envs

    # LPMD per-pair table
    T_pairs.check(T_pairs.run_device(eng, cs, lk, regions=regions), reads, lk)

    # ME / PM
    qd = int(rng.choice([0, 2, 10]))
    T_quartet.check(T_quartet.run_device(eng, cs, mq, qd, regions=regions), reads, mq, qd)

    # MHL
    mk = dict(min_depth=int(rng.choice([0, 1, 5, 10])), min_cpgs=int(rng.choice([1, 2, 4])), min_qual=mq)
    m0 = T_mhl.run_device(eng, cs, mk, regions=regions)
    T_mhl.check(m0, reads, mk)
    # ... with the flush rule looked at per finished row (k_mhl_rowcheck: the host's choice for sparse calls) and per read (position
    # bitmap in the tile kernel), whatever the batch's density: the same rows bit for bit
    for envs in ({"MTH_MHL_ROWCHK": "1"}, {"MTH_MHL_ROWCHK": "0"}, {"MTH_MHL_ROWCHK": "1", "MTH_MHL_FORCE_SUB": "1"},
                 {"MTH_MHL_ROWCHK": "1", "MTH_MHL_TILE_SHIFT": "15"}):
        os.environ.update(envs)
        try:
            m1 = T_mhl.run_device(eng, cs, mk, regions=regions)
        finally:
            for k_ in envs:
                del os.environ[k_]
        assert len(m0["pos"]) == len(m1["pos"]), envs
        assert all((m0[k] == m1[k]).all() for k in ("tid", "pos", "cov")) and (m0["mhl"].view(np.uint32) == m1["mhl"].view(np.uint32)).all(), envs

    # FDRP / qFDRP (a reverse read spanning >= 202 bp can index -1 in the reference and panics there: forward only then)
    fcs, freads = cs, reads
    if read_len > 200:
        from tests import util
        fcs = [util.subset_reads(c, c["read_fwd"] == 1) for c in cs]
        freads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(fcs))
    fk = dict(min_qual=mq, min_depth=int(rng.choice([0, 2, 10])), max_depth=int(rng.choice([2, 8, 40, 64])),
              min_overlap=int(rng.choice([0, 1, 35, 120])), seed=int(rng.integers(0, 1 << 30)))
    fregions = regions
    if regions is not None and fcs is not cs:
        from metheor_amd import shard
        fregions = [[(b, e) for (b, e) in r] for r in regions]
    fdrp_checked(eng, fcs, fk, fregions, freads)


def shift_contig(c, off):
    """the same reads on a contig whose coordinates start `off` bp further right"""
    d = dict(c)
    d["read_start"] = (c["read_start"].astype(np.int64) + off).astype(np.int32)
    d["read_end"] = (c["read_end"].astype(np.int64) + off).astype(np.int32)
    pos = (c["cpg_pos"] & np.uint32(0x7fffffff)).astype(np.int64) + off
    d["cpg_pos"] = (pos.astype(np.uint32) | (c["cpg_pos"] & np.uint32(0x80000000))).astype(np.uint32)
    d["length"] = int(c["length"] + off)
    return d


@pytest.mark.parametrize("seed,top", [(3, 2**31 - 1), (7, 2**31 - 1), (11, 2**30 + 5000), (12, 2**30 - 3000), (21, 2**29 + 77)])
def test_high_coordinates(eng, seed, top):
    """positions up to the top of the int32 range (and around 2^30 / 2^29, where the packed LDS addressing of the tile
    kernel wraps by design): every measure on a region slice [length - span, length) of a contig that long"""
    from metheor_amd import PdrLpmdParams, synth
    rng, cs, _, read_len = scenario(seed)
    c = cs[0]
    off = top - c["length"]
    hc = shift_contig(c, off)
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(hc))
    region = [[(off - 1000, hc["length"])]]          # the batch is a region far from 0 (the index origin sits just below it)
    mq = 10
    pk = dict(min_depth=2, min_cpgs=1, min_qual=mq)
    lk = dict(min_distance=1, max_distance=40, min_qual=mq)
    p = PdrLpmdParams(min_distance=1, max_distance=40, lpmd_min_qual=mq, **pk)
    d, l = T_pdr.run_device(eng, [hc], p, regions=region)
    T_pdr.check_against_oracle(d, l, reads, pk, lk)
    assert len(d["pos"]) == 0 or int(d["pos"].max()) > top - c["length"]
    T_pairs.check(T_pairs.run_device(eng, [hc], lk, regions=region), reads, lk)
    T_quartet.check(T_quartet.run_device(eng, [hc], mq, 1, regions=region), reads, mq, 1)
    mk = dict(min_depth=1, min_cpgs=1, min_qual=mq)
    T_mhl.check(T_mhl.run_device(eng, [hc], mk, regions=region), reads, mk)
    if read_len <= 200:
        fk = dict(min_qual=mq, min_depth=2, max_depth=40, min_overlap=10, seed=1)
        fdrp_checked(eng, [hc], fk, region, reads)
