"""Contig groups (include/metheor_hip.h): several contigs laid out in one virtual coordinate space and accumulated as ONE batch whose
tid is the group's handle; every fetch maps the rows back to (tid, position).  Every measure against the oracle run per contig --
the reference never looks across contigs (is_before(), readutil.rs:304-310) --, bit for bit as in the per-contig tests; through the
C ABI with hand-made offsets (host- and device-resident batches, offsets near 2^31) and through the CLI (mth_decoded_group), where
METHEOR_GROUP=0 must give byte-identical files."""
import os
import subprocess

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metheor_amd", "metheor")
f32 = np.float32


def _contigs(seed, n=5, read_len=150):
    from metheor_amd import synth
    rng = np.random.default_rng(seed)
    return [synth.make_contig(t, int(rng.integers(20_000, 90_000)), int(rng.integers(1_500, 9_000)), float(rng.choice([0.02, 0.05])), rng, read_len=read_len)
            for t in range(n)]


def _group_batch(cs, voff, handle, device):
    """the contigs' arrays concatenated with positions shifted by voff[k]"""
    off, base = [], 0
    for c in cs:
        o = c["cpg_off"].astype(np.int64)
        off.append(o[:-1] + base); base += int(o[-1])
    off.append(np.array([base], np.int64))
    pos = [(((c["cpg_pos"].astype(np.int64) & 0x7fffffff) + v) | (c["cpg_pos"].astype(np.int64) & 0x80000000)).astype(np.uint32) for c, v in zip(cs, voff)]
    g = dict(tid=handle, length=int(voff[-1] + cs[-1]["length"]),
             read_start=np.concatenate([c["read_start"].astype(np.int64) + v for c, v in zip(cs, voff)]).astype(np.int32),
             read_end=np.concatenate([c["read_end"].astype(np.int64) + v for c, v in zip(cs, voff)]).astype(np.int32),
             read_mapq=np.concatenate([c["read_mapq"] for c in cs]), cpg_off=np.concatenate(off).astype(np.uint32),
             cpg_pos=np.concatenate(pos), cpg_rel=np.concatenate([c["cpg_rel"] for c in cs]))
    return util.device_batch(g, device=device)


def _oracle(cs):
    """per contig, under the contig's own tid (the reservoir draw of FDRP / qFDRP is keyed by it)"""
    out = []
    for c in cs:
        r = util.contig_to_records(c, "x")
        refs = [("ctg%d" % t, max(c["length"], 1)) for t in range(c["tid"] + 1)]
        out.append(pyoracle.Reads.decode(bamio.Records(refs, r.tid + c["tid"], r.pos, r.flag, r.mapq, r.cigars, r.xms)))
    return out


def _cat(tabs, tids):
    for t, k in zip(tabs, tids):
        assert (t.tid == k).all()
    return (np.concatenate([np.full(len(t.tid), k, np.int32) for t, k in zip(tabs, tids)]), np.concatenate([t.pos for t in tabs]),
            np.concatenate([t.val for t in tabs]), np.concatenate([t.cnt for t in tabs]))


@pytest.mark.parametrize("device,far", [(None, False), ("cuda:0", False), ("cuda:0", True)])
def test_all_measures_on_a_group(device, far):
    import metheor_amd
    cs = _contigs(11 if not far else 12)
    # offsets: gaps of span + ~1000 like mth_decoded_group makes; far: the last contigs up against 2^31
    voff, v = [], 0
    for c in cs:
        voff.append(v); v += ((c["length"] + 150 + 1024 + 4095) // 4096) * 4096
    if far:
        top = (1 << 31) - (1 << 22) - cs[-1]["length"] - 8192
        voff[-1] = top // 4096 * 4096
        voff[-2] = (top - 700_000_000) // 4096 * 4096
    tids = [3, 5, 6, 20, 21]                                  # ascending, not 0..n-1
    for c, t in zip(cs, tids):
        c["tid"] = t
    ora = _oracle(cs)
    e = metheor_amd.Engine(0)
    try:
        h = e.group_define(tids, voff)
        assert h <= -2
        bt = _group_batch(cs, voff, h, device)
        # PDR + LPMD
        e.reset(); e.pdr_lpmd_accumulate(bt, metheor_amd.PdrLpmdParams(min_depth=3, min_cpgs=2))
        d = e.pdr_fetch()
        t_, p_, v_, c_ = _cat([o.pdr(min_depth=3, min_cpgs=2) for o in ora], tids)
        assert (d["tid"] == t_).all() and (d["pos"] == p_[:, 0]).all() and (d["pdr"].view(np.uint32) == v_.view(np.uint32)).all()
        assert (d["n_concordant"] == c_[:, 0]).all() and (d["n_discordant"] == c_[:, 1]).all()
        lg = e.lpmd_global()
        assert lg["n_concordant"] == sum(o.lpmd()["n_concordant"] for o in ora) and lg["n_discordant"] == sum(o.lpmd()["n_discordant"] for o in ora)
        # MHL
        e.reset(); e.mhl_accumulate(bt, min_depth=3, min_cpgs=2)
        d = e.mhl_fetch()
        t_, p_, v_, c_ = _cat([o.mhl(min_depth=3, min_cpgs=2) for o in ora], tids)
        assert (d["tid"] == t_).all() and (d["pos"] == p_[:, 0]).all() and (d["mhl"].view(np.uint32) == v_.view(np.uint32)).all()
        # FDRP / qFDRP without and with the reservoir (the draw is keyed by the REAL tid and position)
        for D, seed in ((40, 0), (4, 9)):
            e.reset(); e.fdrp_accumulate(bt, min_depth=2, max_depth=D, seed=seed)
            d = e.fdrp_fetch()
            t_, p_, v_, c_ = _cat([o.fdrp(min_depth=2, max_depth=D, seed=seed) for o in ora], tids)
            q_ = np.concatenate([o.qfdrp(min_depth=2, max_depth=D, seed=seed).val for o in ora])
            assert (d["tid"] == t_).all() and (d["pos"] == p_[:, 0]).all(), D
            assert (d["fdrp"].view(np.uint32) == v_.view(np.uint32)).all() and (d["qfdrp"].view(np.uint32) == q_.view(np.uint32)).all(), D
        # ME / PM (rows: (tid, pos1..pos4) order per batch = the group's order)
        e.reset(); e.quartet_accumulate(bt)
        d = e.quartet_fetch(min_depth=2)
        t_, p_, v_, c_ = _cat([o.pm(min_depth=2) for o in ora], tids)
        m_ = np.concatenate([o.me(min_depth=2).val for o in ora])
        order = np.lexsort((d["pos"][:, 3], d["pos"][:, 2], d["pos"][:, 1], d["pos"][:, 0], d["tid"]))
        assert (d["tid"][order] == t_).all() and (d["pos"][order] == p_).all() and (d["cnt"][order] == c_).all()
        assert (d["pm"][order].view(np.uint32) == v_.view(np.uint32)).all() and np.abs(d["me"][order].astype(np.float64) - m_).max() <= 1e-6
        # the pairs table
        e.reset(); e.lpmd_pairs_accumulate(bt)
        d = e.lpmd_pairs_fetch()
        t_, p_, v_, c_ = _cat([o.lpmd(pairs=True)["pairs"] for o in ora], tids)
        assert (d["tid"] == t_).all() and (d["pos1"] == p_[:, 0]).all() and (d["pos2"] == p_[:, 1]).all()
        assert (d["n_concordant"] == c_[:, 0]).all() and (d["lpmd"].view(np.uint32) == v_.view(np.uint32)).all()
        # a plain batch and a group in one job; count-only and tid-only fetches
        extra = _contigs(13, n=1)[0]; extra["tid"] = 30
        e.reset(); e.mhl_accumulate(bt, min_depth=3, min_cpgs=2); e.mhl_accumulate(util.device_batch(extra, device=device), min_depth=3, min_cpgs=2)
        d = e.mhl_fetch()
        t2 = _oracle([extra])[0].mhl(min_depth=3, min_cpgs=2)
        t_, p_, v_, c_ = _cat([o.mhl(min_depth=3, min_cpgs=2) for o in ora] + [t2], tids + [30])
        assert (d["tid"] == t_).all() and (d["pos"] == p_[:, 0]).all()
    finally:
        e.close()


def test_group_define_is_checked():
    import metheor_amd
    e = metheor_amd.Engine(0)
    try:
        for tids, voff in (([0, 1], [0, 0]), ([0, 1], [4096, 0]), ([0, -1], [0, 4096]), ([0], [-5]), ([0, 1], [0, 1 << 31])):
            with pytest.raises(metheor_amd.MthError):
                e.group_define(tids, voff)
        assert e.group_define([0, 1], [0, 4096]) == -2 and e.group_define([2, 7], [0, 1 << 20]) == -3
        e.group_clear()
        assert e.group_define([0, 1], [0, 4096]) == -2
    finally:
        e.close()


# ---- through the CLI: mth_decoded_group packs the decoded stream's contigs ------------------------------------------------------------
def _run(env, *args):
    return subprocess.run([EXE, *[str(a) for a in args]], capture_output=True, text=True, cwd=ROOT, timeout=900, env=dict(os.environ, **env))


def _bam(tmp_path, n_contigs, seed, reads=(40, 400), length=(2_000, 9_000), read_len=150):
    from metheor_amd import synth
    rng = np.random.default_rng(seed)
    cs = [synth.make_contig(t, int(rng.integers(*length)), int(rng.integers(*reads)), 0.06, rng, read_len=read_len) for t in range(n_contigs)]
    recs = [util.contig_to_records(c, "ctg%d" % t) for t, c in enumerate(cs)]
    rec = bamio.Records([r.refs[0] for r in recs], np.concatenate([r.tid + t for t, r in enumerate(recs)]), np.concatenate([r.pos for r in recs]),
                        np.concatenate([r.flag for r in recs]), np.concatenate([r.mapq for r in recs]), sum((r.cigars for r in recs), []), sum((r.xms for r in recs), []))
    bam = str(tmp_path / "g.bam")
    bamio.write_bam(bam, rec)
    return bam, rec


@pytest.mark.parametrize("n_contigs,seed,read_len", [(6, 1, 150), (400, 2, 150), (7, 3, 230)])      # 230: the exact PDR walk, FDRP's general walk
def test_cli_groups_equal_one_batch_per_contig(tmp_path, n_contigs, seed, read_len):
    bam, rec = _bam(tmp_path, n_contigs, seed, read_len=read_len)
    reads = pyoracle.Reads.decode(rec)
    names = [r[0] for r in rec.refs]
    for sub, extra in (("pdr", ["-d", 2, "-p", 2]), ("mhl", ["-d", 2, "-p", 2]), ("fdrp", ["-d", 2, "-D", 5]), ("qfdrp", ["-d", 2]),
                       ("me", ["-d", 2]), ("pm", ["-d", 2]), ("lpmd", ["-p", tmp_path / "pairs.tsv"])):
        outs = []
        for grp in ("1", "0"):
            o = tmp_path / ("o%s.tsv" % grp)
            r = _run({"METHEOR_GROUP": grp, "METHEOR_TIMING": "1"}, sub, "-i", bam, "-o", o, *extra)
            if r.returncode == 101 and "fdrp.rs:70-72" in r.stderr and sub in ("fdrp", "qfdrp"):      # reads beyond 201 bases can hit the reference's own panic: both ways then
                outs.append("panic")
                continue
            assert r.returncode == 0, (sub, grp, r.stderr)
            assert ("contig groups" in r.stderr) == (grp == "1"), r.stderr
            outs.append(o.read_text() + ((tmp_path / "pairs.tsv").read_text() if sub == "lpmd" else ""))
        if sub in ("me", "pm"):
            assert sorted(outs[0].splitlines()) == sorted(outs[1].splitlines()), sub          # unsorted output (HashMap order in the reference)
        else:
            assert outs[0] == outs[1], sub
        if sub == "pdr":
            assert outs[0] == util.oracle_tsv_pdr(reads, names, min_depth=2, min_cpgs=2, min_qual=10)
        if sub == "mhl":
            t = reads.mhl(min_depth=2, min_cpgs=2)
            assert outs[0] == "".join("%s\t%d\t%d\t%s\n" % (names[ti], p, p + 2, pyoracle.format_f32(v)) for ti, p, v in zip(t.tid, t.pos[:, 0], t.val))


def test_cli_several_groups_and_single_contig_groups(tmp_path):
    """MTH_GROUP_MAX_POSITIONS (what free device memory does on a shared GPU) cuts the packing short: several groups, some of one contig"""
    bam, rec = _bam(tmp_path, 9, 5)
    reads = pyoracle.Reads.decode(rec)
    names = [r[0] for r in rec.refs]
    for cap in (12_000, 30_000):
        o = tmp_path / "o.tsv"
        r = _run({"MTH_GROUP_MAX_POSITIONS": str(cap)}, "pdr", "-i", bam, "-o", o, "-d", 2, "-p", 2)
        assert r.returncode == 0, r.stderr
        assert o.read_text() == util.oracle_tsv_pdr(reads, names, min_depth=2, min_cpgs=2, min_qual=10)
        r = _run({"MTH_GROUP_MAX_POSITIONS": str(cap)}, "fdrp", "-i", bam, "-o", o, "-d", 2, "-D", 6)
        assert r.returncode == 0, r.stderr
        t = reads.fdrp(min_depth=2, max_depth=6)
        assert o.read_text() == "".join("%s\t%d\t%d\t%s\n" % (names[ti], p, p + 2, pyoracle.format_f32(v)) for ti, p, v in zip(t.tid, t.pos[:, 0], t.val))
