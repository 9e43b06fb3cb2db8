"""GPU parity for the BASELINE.json configurations that had no driver-run test:

config 3  "Synthetic 200M-read WGBS (150bp, 30x human), all 7 measures, 1xMI355X" -- the 24 hg38-sized contigs at a size
          the oracle finishes in seconds (every measure + the pairs table, bit-exact / 1e-6), and one chr1-sized contig at
          the configuration's FULL density and depth through size-independent properties (closed forms on the SoA,
          idempotence, an oracle-checked prefix);
config 5  "Whole-genome 1B-read WGBS sharded by contig across 8xMI355X with RCCL final reduce" -- the same generator written
          as ONE 24-contig BAM, `metheor <measure> --gpus 8` (eight shards, eight device contexts, the RCCL/all-reduce entry
          point for LPMD; all on the one GPU a test box has), every measure against the ORACLE's text.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle
from tests import test_gpu_fdrp as T_fdrp
from tests import test_gpu_mhl as T_mhl
from tests import test_gpu_pairs as T_pairs
from tests import test_gpu_pdr_lpmd as T_pdr
from tests import test_gpu_quartet as T_quartet
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metheor_amd", "metheor")


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def wgbs24():
    """S-WGBS scaled to 1.5 M reads: 24 contigs, hg38 lengths, dense windows (~60x) inside an almost empty genome"""
    from metheor_amd import synth
    cs = synth.wgbs_small(1_500_000)
    reads = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs))
    return cs, reads


def test_config3_wgbs_all_measures_vs_oracle(eng, wgbs24):
    from metheor_amd import PdrLpmdParams, synth
    cs, reads = wgbs24
    assert len(cs) == 24 and cs[0]["length"] == synth.HG38_LENGTHS[0] and sum(len(c["read_start"]) for c in cs) > 1_400_000
    # reference CLI defaults (lib.rs:36-46, 206-214) and a permissive set that also emits the sparse part's rows
    for pk, lk in ((dict(min_depth=10, min_cpgs=4, min_qual=10), dict(min_distance=2, max_distance=16, min_qual=10)),
                   (dict(min_depth=1, min_cpgs=1, min_qual=0), dict(min_distance=1, max_distance=40, min_qual=0))):
        p = PdrLpmdParams(min_distance=lk["min_distance"], max_distance=lk["max_distance"], lpmd_min_qual=lk["min_qual"], **pk)
        d, l = T_pdr.run_device(eng, cs, p, device="cuda:0")
        T_pdr.check_against_oracle(d, l, reads, pk, lk)
        assert len(d["pos"]) > 1000 and len(set(d["tid"].tolist())) == 24
    lk = dict(min_distance=2, max_distance=16, min_qual=10)
    T_pairs.check(T_pairs.run_device(eng, cs, lk), reads, lk)
    for mq, qd in ((10, 10), (0, 1)):
        T_quartet.check(T_quartet.run_device(eng, cs, mq, qd), reads, mq, qd)
    for mk in (dict(min_depth=10, min_cpgs=4, min_qual=10), dict(min_depth=1, min_cpgs=1, min_qual=10)):
        T_mhl.check(T_mhl.run_device(eng, cs, mk), reads, mk)
    # -D 64 with ~60x windows: some sites sample (device and oracle share the counter-based draw), most do not
    for fk in (dict(min_qual=10, min_depth=10, max_depth=64, min_overlap=35, seed=3), dict(min_qual=10, min_depth=2, max_depth=40, min_overlap=35, seed=5)):
        T_fdrp.check(T_fdrp.run_device(eng, cs, fk), reads, fk)


def test_config3_full_density_chr1_properties(eng):
    """one chr1-sized contig at S-WGBS-200M's own density (0.0091) and depth (200 M x 248.96 / 3088 Mbp = 16.1 M reads, 9.7x):
    the tile count (60 781), per-tile occupancy and call density of the real configuration, checked through properties"""
    from metheor_amd import PdrLpmdParams, shard, synth
    ln = synth.HG38_LENGTHS[0]
    n = int(round(200_000_000 * ln / float(sum(synth.HG38_LENGTHS))))
    c = synth.make_contig(0, ln, n, 0.0091, np.random.default_rng(2000))
    bt = util.device_batch(c, device="cuda:0")
    kw = dict(min_depth=0, min_cpgs=0, min_qual=10)
    eng.reset()
    eng.pdr_lpmd_accumulate(bt, PdrLpmdParams(**kw))
    p, l = eng.pdr_fetch(), eng.lpmd_global()
    off = c["cpg_off"].astype(np.int64)
    ncpg = np.diff(off)
    assert 1.2 < ncpg.mean() < 1.5
    ok = (c["read_mapq"] >= 10) & (ncpg > 0)
    meth = (c["cpg_pos"] >> 31).astype(np.int64)
    msum = np.add.reduceat(np.concatenate([meth, [0]]), off[:-1])
    msum[ncpg == 0] = 0
    disc = (msum > 0) & (msum < ncpg)
    assert int(p["n_concordant"].sum()) == int(ncpg[ok & ~disc].sum())
    assert int(p["n_discordant"].sum()) == int(ncpg[ok & disc].sum())
    called = np.unique((c["cpg_pos"] & 0x7fffffff)[np.repeat(ok, ncpg)])
    assert len(p["pos"]) == len(called) and (p["pos"] == called).all()
    assert l["n_read"] == n and l["n_valid_read"] == int((c["read_mapq"] >= 10).sum())
    # LPMD pair counts in closed form for the default window: pairs (j < k) of a read with 2 <= rel_k - rel_j <= 16
    rel = c["cpg_rel"].astype(np.int64)
    lp_ok = np.repeat(c["read_mapq"] >= 10, ncpg)
    read_of = np.repeat(np.arange(n), ncpg)
    tot_c = tot_d = 0
    for g in range(1, 9):                              # CpGs are >= 2 bp apart: at most 8 gaps inside 16 bp
        same = (read_of[g:] == read_of[:-g]) & lp_ok[g:]
        dist = rel[g:] - rel[:-g]
        inw = same & (dist >= 2) & (dist <= 16)
        dd = inw & (meth[g:] != meth[:-g])
        tot_d += int(dd.sum()); tot_c += int(inw.sum()) - int(dd.sum())
    assert l["n_concordant"] == tot_c and l["n_discordant"] == tot_d
    # the same resident batch through the other tile measures: mass conservation
    eng.quartet_accumulate(bt, 10)
    q = eng.quartet_fetch(0)
    assert int(q["cnt"].sum()) == int(np.maximum(ncpg[c["read_mapq"] >= 10] - 3, 0).sum())
    eng.lpmd_pairs_accumulate(bt, 2, 16, 10)
    pr = eng.lpmd_pairs_fetch()
    assert int(pr["n_concordant"].sum()) == tot_c and int(pr["n_discordant"].sum()) == tot_d
    # idempotence
    eng.reset()
    eng.pdr_lpmd_accumulate(bt, PdrLpmdParams(**kw))
    p2 = eng.pdr_fetch()
    assert all((p[k] == p2[k]).all() for k in p)
    # exact against the oracle on a 400 k-read prefix and on a 400 k-read slice at the contig's far end
    slices = []
    for lo_i, hi_i in ((0, 400_000), (n - 400_000, n)):
        beg = 0 if lo_i == 0 else int(c["read_start"][lo_i]) + 200
        end = int(c["read_start"][hi_i]) if hi_i < n else ln
        sub = shard.slice_region(c, max(beg - 400, 0), end)
        reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(sub))
        o = reads.pdr(**kw)
        keep = (o.pos[:, 0] >= beg) & (o.pos[:, 0] < end)
        m = (p["pos"] >= beg) & (p["pos"] < end)
        assert m.sum() > 30_000 and (p["pos"][m] == o.pos[keep, 0]).all()
        assert (p["n_concordant"][m] == o.cnt[keep, 0]).all() and (p["n_discordant"][m] == o.cnt[keep, 1]).all()
        assert (p["pdr"][m].view(np.uint32) == o.val[keep].view(np.uint32)).all()
        slices.append((beg, end, reads))
    # MHL and FDRP / qFDRP over the whole 16.1 M-read contig, exact against the oracle on the same two slices:
    # a site in [beg, end) only ever sees reads that start within a read length of it, all of them in the slice
    mk = dict(min_depth=10, min_cpgs=4, min_qual=10)
    eng.reset()
    eng.mhl_accumulate(bt, **mk)
    dm = eng.mhl_fetch()
    fk = dict(min_qual=10, min_depth=10, max_depth=40, min_overlap=35)
    eng.fdrp_accumulate(bt, **fk)
    df = eng.fdrp_fetch()
    assert len(dm["pos"]) > 20_000 and len(df["pos"]) > 100_000      # (MHL: reads with >= 4 CpGs are rare at 1.35 calls per read)
    for beg, end, reads in slices:
        o = reads.mhl(**mk)
        keep = (o.pos[:, 0] >= beg) & (o.pos[:, 0] < end)
        m = (dm["pos"] >= beg) & (dm["pos"] < end)
        assert m.sum() > 300 and (dm["pos"][m] == o.pos[keep, 0]).all() and (dm["cov"][m] == o.cnt[keep, 0]).all()
        assert np.abs(dm["mhl"][m].astype(np.float64) - o.val[keep].astype(np.float64)).max() <= 1e-6
        of, oq = reads.fdrp(**fk), reads.qfdrp(**fk)
        keep = (of.pos[:, 0] >= beg) & (of.pos[:, 0] < end)
        m = (df["pos"] >= beg) & (df["pos"] < end)
        assert m.sum() > 5_000 and (df["pos"][m] == of.pos[keep, 0]).all() and (df["n_reads"][m] == of.cnt[keep, 0]).all()
        assert T_fdrp.same(df["fdrp"][m], of.val[keep]) <= 1e-6 and T_fdrp.same(df["qfdrp"][m], oq.val[keep]) <= 1e-6


def test_config4_full_size_hotspots(eng):
    """BASELINE config 4 at its full size (20 000 windows, 6.66 M reads, 1.40 M sites, 1.8e9 read pairs), -D 64 and the CLI's
    default -D 40: rows and stored-read counts in closed form from the SoA (a site's depth = the calls it gets from mapq-passing
    reads; every 150-bp read fits the +-201-bp window of fdrp.rs:58-63), values exact against the oracle on two 30-window slices"""
    from metheor_amd import shard, synth
    c = synth.hotspots()                                   # n_windows 20000, depth 50, density 0.08, seed 50
    n = len(c["read_start"])
    assert n == 6_660_000
    bt = util.device_batch(c, device="cuda:0")
    ncpg = np.diff(c["cpg_off"].astype(np.int64))
    ok = np.repeat(c["read_mapq"] >= 10, ncpg)
    pos_ok = (c["cpg_pos"] & 0x7fffffff)[ok].astype(np.int64)
    sites, depth = np.unique(pos_ok, return_counts=True)
    want = depth >= 10
    out = {}
    for D in (64, 40):
        fk = dict(min_qual=10, min_depth=10, max_depth=D, min_overlap=35, seed=4)
        eng.reset()
        eng.fdrp_accumulate(bt, **fk)
        d = eng.fdrp_fetch()
        # (a site flushed by a read that starts on its G and re-opened by a reverse read with the same start reports its LAST
        # segment, SURVEY Q1: a fraction of a percent of the sites hold fewer reads than their depth, or miss min_depth)
        idx = np.searchsorted(sites, d["pos"])
        assert (np.diff(d["pos"]) > 0).all() and (sites[idx] == d["pos"]).all() and want[idx].all()
        assert len(d["pos"]) >= 0.99 * int(want.sum())
        cap = np.minimum(depth[idx], D)
        assert (d["n_reads"] <= cap).all() and (d["n_reads"] == cap).mean() > 0.98      # measured: 98.8 % at this density and depth
        assert (d["fdrp"] >= 0).all() and (d["fdrp"] <= 1).all() and (d["qfdrp"] >= 0).all() and (d["qfdrp"] <= 1).all()
        out[D] = d
    stride = 1000 + 2 * 150 + 404
    for w0 in (0, 12_345):
        beg, end = w0 * stride, (w0 + 30) * stride
        sub = shard.slice_region(c, beg, end)
        reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(sub))
        for D in (64, 40):
            fk = dict(min_qual=10, min_depth=10, max_depth=D, min_overlap=35, seed=4)
            of, oq = reads.fdrp(**fk), reads.qfdrp(**fk)
            m = (out[D]["pos"] >= beg) & (out[D]["pos"] < end)
            assert m.sum() == len(of) > 1500 and (out[D]["pos"][m] == of.pos[:, 0]).all()
            assert T_fdrp.same(out[D]["fdrp"][m], of.val) <= 1e-6 and T_fdrp.same(out[D]["qfdrp"][m], oq.val) <= 1e-6


def test_config5_sharded_8_vs_oracle_text(wgbs24, tmp_path):
    """one 24-contig BAM, `metheor <sub> --gpus 8`: eight shards cut by compressed size (inside contigs and at contig
    changes), eight device contexts, LPMD counters through mth_allreduce_lpmd -- every output file against the ORACLE's
    text (not against another run of the same engine)"""
    from metheor_amd import hostapi, synth
    cs, reads = wgbs24
    bam = str(tmp_path / "wgbs24.bam")
    hostapi.write_synthetic_bam_multi(bam, cs, synth.HG38_NAMES, seed=11)
    env = dict(os.environ, METHEOR_SEED="9")
    cases = (("pdr", []), ("pdr", ["-d", "1", "-p", "1"]), ("lpmd", []), ("mhl", ["-d", "5", "-p", "2"]), ("me", ["-d", "5"]), ("pm", ["-d", "5"]),
             ("fdrp", ["-d", "5", "-D", "64"]), ("qfdrp", ["-d", "5", "-D", "64"]))
    for sub, extra in cases:
        o, pf = tmp_path / ("%s.tsv" % sub), tmp_path / "pairs.tsv"
        args = [EXE, sub, "-i", bam, "-o", str(o)] + extra + ["--gpus", "8"]
        if sub == "lpmd":
            args += ["-p", str(pf)]
        r = subprocess.run(args, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
        assert r.returncode == 0, (sub, r.stderr)
        want, want_pairs = util.oracle_text(reads, synth.HG38_NAMES, sub, input_name=bam, seed=9, **util.oracle_kwargs(sub, extra))
        util.assert_tsv_equals_oracle(sub, o.read_text(), want)
        assert len(want) > (20 if sub == "lpmd" else 10_000), sub
        if sub == "lpmd":
            assert pf.read_text() == want_pairs and want_pairs.count("\n") > 10_000


def test_config3_full_size_200M_reads_properties():
    """BASELINE config 3 AT FULL SIZE under pytest: S-WGBS-200M -- 200 M x 150-bp reads over the 24 hg38-sized contigs,
    generated on the device (metheor_amd/synth_device.py) -- every measure over every contig, checked through size-independent
    properties computed from the SoA with torch on the device, per contig: PDR's per-strand mass (sum of n_concordant / n_discordant
    = calls of the concordant / discordant passing reads) and its site list (= the distinct positions those reads call); LPMD's four
    counters in closed form (the pair count of the default window by shifted comparisons); the quartet histograms' mass and the
    pairs table's column sums; MHL / FDRP / qFDRP: the rows are exactly the sites with >= min_depth contributing reads, minus the
    few whose contributors fall into several segments (bounded below by 0.98 of them), coverage <= contributors, values in [0, 1]."""
    import torch
    import metheor_amd
    from metheor_amd import PdrLpmdParams, synth_device
    dev = torch.device("cuda", 0)
    eng = metheor_amd.Engine(0)
    kw = dict(min_depth=10, min_cpgs=4, min_qual=10)
    tot_reads = 0
    exp = dict(nc=0, nd=0, n_read=0, n_valid=0, lc=0, ld=0, quart=0)
    sites_pdr, rows = [], dict(mhl=0, fdrp=0, mhl_max=0, fdrp_max=0)
    eng.reset()
    for bt, info in synth_device.wgbs(n_reads=200_000_000, device=dev):
        rs, mq, off, pos, rel = bt.keep[0], bt.keep[2], bt.keep[3].to(torch.int64) & 0xffffffff, bt.keep[4], bt.keep[5]
        n = int(rs.shape[0])
        tot_reads += n
        ncpg = off[1:] - off[:-1]
        total = int(off[-1].item())
        read_of = torch.repeat_interleave(torch.arange(n, device=dev), ncpg, output_size=total)
        meth = ((pos.to(torch.int64) >> 31) & 1)
        p31 = pos.to(torch.int64) & 0x7fffffff
        msum = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, read_of, meth)
        disc = (msum > 0) & (msum < ncpg)
        okq = mq >= 10
        ok_pdr = okq & (ncpg >= kw["min_cpgs"])
        # PDR (pdr.rs:147-191): every call of a passing read adds one to its site
        cov = torch.zeros(info["length"] + 2, dtype=torch.int32, device=dev)
        cov.index_add_(0, p31[ok_pdr[read_of]], torch.ones(int(ok_pdr[read_of].sum().item()), dtype=torch.int32, device=dev))
        keep_site = cov >= kw["min_depth"]
        dsc = torch.zeros(info["length"] + 2, dtype=torch.int32, device=dev)
        sel = (ok_pdr & disc)[read_of]
        dsc.index_add_(0, p31[sel], torch.ones(int(sel.sum().item()), dtype=torch.int32, device=dev))
        exp["nd"] += int(dsc[keep_site].sum().item())
        exp["nc"] += int((cov - dsc)[keep_site].sum().item())
        sites_pdr.append(torch.nonzero(keep_site).flatten().to(torch.int32).cpu().numpy())
        # LPMD (lpmd.rs:176-190)
        exp["n_read"] += n
        exp["n_valid"] += int(okq.sum().item())
        lp_ok = okq[read_of]
        r64 = rel.to(torch.int64)
        for g in range(1, 9):                           # CpGs are >= 2 bp apart: at most 8 gaps inside 16 bp
            same = (read_of[g:] == read_of[:-g]) & lp_ok[g:]
            dist = r64[g:] - r64[:-g]
            inw = same & (dist >= 2) & (dist <= 16)
            dd = inw & (meth[g:] != meth[:-g])
            d = int(dd.sum().item())
            exp["ld"] += d
            exp["lc"] += int(inw.sum().item()) - d
        exp["quart"] += int(torch.clamp(ncpg[okq] - 3, min=0).sum().item())
        # MHL: contributors = passing reads with >= min_cpgs calls (mhl.rs:176-190); FDRP: passing reads with >= 1 call (fdrp.rs:205-231)
        rows["mhl_max"] += int((cov >= kw["min_depth"]).sum().item())
        covf = torch.zeros(info["length"] + 2, dtype=torch.int32, device=dev)
        covf.index_add_(0, p31[lp_ok], torch.ones(int(lp_ok.sum().item()), dtype=torch.int32, device=dev))
        rows["fdrp_max"] += int((covf >= 10).sum().item())
        del cov, dsc, covf, read_of, meth, p31, msum, r64
        eng.pdr_lpmd_accumulate(bt, PdrLpmdParams(**kw))
        eng.quartet_accumulate(bt, 10)
        eng.lpmd_pairs_accumulate(bt, 2, 16, 10)
        eng.mhl_accumulate(bt, **kw)
        eng.fdrp_accumulate(bt, min_qual=10, min_depth=10, max_depth=40, min_overlap=35)
        eng.sync()                                      # the batch's tensors go away with the loop variable
    assert tot_reads == 200_000_000 or abs(tot_reads - 200_000_000) < 24
    p, l = eng.pdr_fetch(), eng.lpmd_global()
    want_pos = np.concatenate(sites_pdr)
    assert len(p["pos"]) == len(want_pos) and (p["pos"] == want_pos).all()
    assert (np.diff(p["tid"]) >= 0).all() and p["tid"][0] == 0 and p["tid"][-1] == 23
    assert int(p["n_concordant"].astype(np.int64).sum()) == exp["nc"] and int(p["n_discordant"].astype(np.int64).sum()) == exp["nd"]
    assert ((p["n_concordant"] + p["n_discordant"]) >= kw["min_depth"]).all()
    assert (l["n_read"], l["n_valid_read"], l["n_concordant"], l["n_discordant"]) == (exp["n_read"], exp["n_valid"], exp["lc"], exp["ld"])
    q = eng.quartet_fetch(0)
    assert int(q["cnt"].astype(np.int64).sum()) == exp["quart"]
    pr = eng.lpmd_pairs_fetch()
    assert int(pr["n_concordant"].astype(np.int64).sum()) == exp["lc"] and int(pr["n_discordant"].astype(np.int64).sum()) == exp["ld"]
    dm, df = eng.mhl_fetch(), eng.fdrp_fetch()
    assert 0.98 * rows["mhl_max"] <= len(dm["pos"]) <= rows["mhl_max"], (len(dm["pos"]), rows)
    assert 0.98 * rows["fdrp_max"] <= len(df["pos"]) <= rows["fdrp_max"], (len(df["pos"]), rows)
    assert (dm["cov"] >= 10).all() and np.nanmin(dm["mhl"]) >= 0.0 and np.nanmax(dm["mhl"]) <= 1.0
    assert (df["n_reads"] >= 10).all() and (df["n_reads"] <= 40).all()
    for k in ("fdrp", "qfdrp"):
        assert np.nanmin(df[k]) >= 0.0 and np.nanmax(df[k]) <= 1.0
    assert len(dm["pos"]) > 300_000 and len(df["pos"]) > 10_000_000
    eng.close()
