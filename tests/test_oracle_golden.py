"""Pins the CPU oracle against EVERY known-answer test the reference holds for the hot path
(SURVEY.md section 8c).  Each case cites the reference test it restates; inputs are the
reference's own BAM fixtures (tests/golden/test{1..6}.bam, copied data files)."""
import os

import numpy as np
import pytest

from oracle import bamio, pyoracle


@pytest.fixture(scope="module")
def reads(golden_dir):
    out = {}
    for k in range(1, 7):
        out[k] = pyoracle.Reads.decode(bamio.read_bam(os.path.join(golden_dir, "test%d.bam" % k)))
    return out


f32 = np.float32

# ---------------------------------------------------------------- readutil.rs:420-440
def test_readutil_test1_discordant_reads(reads):
    r = reads[1]
    assert len(r) == 16
    assert sum(r.is_discordant(i) for i in range(16)) == 14


# ---------------------------------------------------------------- pdr.rs:218-366
@pytest.mark.parametrize("k,nsite,pdr,nc,nd", [
    (1, 4, 14.0 / 16.0, 2, 14),   # pdr.rs:226-237
    (2, 4, 0.0, 16, 0),           # pdr.rs:248-259
    (3, 4, 0.0, 2, 0),            # pdr.rs:271-282
    (4, 8, 14.0 / 16.0, 2, 14),   # pdr.rs:296-307
])
def test_pdr(reads, k, nsite, pdr, nc, nd):
    t = reads[k].pdr(min_depth=0, min_cpgs=0, min_qual=10)
    assert len(t) == nsite
    assert (t.val == f32(pdr)).all()
    assert (t.cnt[:, 0] == nc).all() and (t.cnt[:, 1] == nd).all()


def test_pdr_positions(reads):
    assert reads[1].pdr(0, 0, 10).pos[:, 0].tolist() == [0, 2, 4, 6]
    assert reads[4].pdr(0, 0, 10).pos[:, 0].tolist() == [0, 2, 4, 6, 13, 15, 17, 19]


def test_pdr_test5_empty(reads):  # pdr.rs:311-323
    assert len(reads[5].pdr(0, 0, 10)) == 0


def test_pdr_test6(reads):  # pdr.rs:326-365
    t = reads[6].pdr(min_depth=0, min_cpgs=1, min_qual=10)
    assert len(t) == 2 and t.pos[:, 0].tolist() == [2, 13]
    assert (t.val == 0).all() and (t.cnt[:, 0] == 16).all() and (t.cnt[:, 1] == 0).all()
    assert len(reads[6].pdr(min_depth=0, min_cpgs=2, min_qual=10)) == 0


# ---------------------------------------------------------------- lpmd.rs:208-269
@pytest.mark.parametrize("k,want", [(1, 0.5), (2, 0.0), (3, 0.0), (4, 0.5)])
def test_lpmd(reads, k, want):
    assert reads[k].lpmd(2, 16, 10)["lpmd"] == f32(want)


def test_lpmd_test1_counts(reads):  # c=48, d=48
    r = reads[1].lpmd(2, 16, 10)
    assert (r["n_concordant"], r["n_discordant"], r["n_read"], r["n_valid_read"]) == (48, 48, 16, 16)


def test_lpmd_test5_nan(reads):  # lpmd.rs:258-269
    assert np.isnan(reads[5].lpmd(2, 16, 10)["lpmd"])


# ---------------------------------------------------------------- mhl.rs:243-295 (startup(): no flush, q10)
@pytest.mark.parametrize("k,want", [(1, 0.1625), (2, 0.5), (3, 0.5), (4, 0.1625)])
def test_mhl(reads, k, want):
    t = reads[k].mhl(min_depth=0, min_cpgs=0, min_qual=10)
    assert len(t) == (8 if k == 4 else 4)
    assert (t.val == f32(want)).all()


def test_mhl_test5_empty(reads):
    assert len(reads[5].mhl(0, 0, 10)) == 0


# ---------------------------------------------------------------- me.rs:138-207 / pm.rs:134-202
@pytest.mark.parametrize("k,nq,me,pm", [
    (1, 1, 1.0, 0.9375), (2, 1, 0.25, 0.5), (3, 1, 0.25, 0.5), (4, 2, 1.0, 0.9375)])
def test_me_pm(reads, k, nq, me, pm):
    tm = reads[k].me(min_depth=0, min_qual=10)
    tp = reads[k].pm(min_depth=0, min_qual=10)
    assert len(tm) == nq and len(tp) == nq
    assert (tm.val == f32(me)).all()
    assert (tp.val == f32(pm)).all()
    if k == 1:
        assert tm.cnt.sum() == 16 and (tm.cnt == 1).all()  # me.rs:149 depth 16, all 16 patterns once
        assert tm.pos.tolist() == [[0, 2, 4, 6]]


def test_me_pm_test5_empty(reads):
    assert len(reads[5].me(0, 10)) == 0 and len(reads[5].pm(0, 10)) == 0


# ---------------------------------------------------------------- fdrp.rs:252-332
def test_fdrp(reads):
    t = reads[1].fdrp(min_qual=0, min_depth=2, max_depth=40, min_overlap=4)   # fdrp.rs:253-268
    assert t.pos[:, 0].tolist() == [0, 2, 4, 6] and (t.val == 1.0).all()
    t = reads[2].fdrp(0, 2, 40, 4)                                            # fdrp.rs:270-285
    assert t.pos[:, 0].tolist() == [0, 2, 4, 6]
    assert (np.abs(t.val - (1.0 - 56.0 / 120.0)) < 1e-4).all()
    t = reads[3].fdrp(1, 2, 40, 4)                                            # fdrp.rs:287-302
    assert t.pos[:, 0].tolist() == [0, 2, 4, 6] and (t.val == 1.0).all()
    t = reads[4].fdrp(1, 2, 40, 4)                                            # fdrp.rs:304-319
    assert t.pos[:, 0].tolist() == [0, 2, 4, 6, 13, 15, 17, 19] and (t.val == 1.0).all()
    assert len(reads[5].fdrp(1, 2, 40, 4)) == 0                               # fdrp.rs:321-332


# ---------------------------------------------------------------- qfdrp.rs:269-439
def test_qfdrp(reads):
    t = reads[1].qfdrp(0, 2, 40, 4)                                           # qfdrp.rs:357-374
    assert t.pos[:, 0].tolist() == [0, 2, 4, 6]
    assert (np.abs(t.val - f32(8.0 / 15.0)) < 1e-5).all()
    assert (t.cnt[:, 0] == 16).all()                                          # qfdrp.rs:309-332
    t = reads[2].qfdrp(0, 2, 40, 4)                                           # qfdrp.rs:376-392 (exact)
    assert (t.val == f32(8.0 / 15.0)).all()
    t = reads[3].qfdrp(1, 2, 40, 4)                                           # qfdrp.rs:394-409
    assert t.pos[:, 0].tolist() == [0, 2, 4, 6] and (t.val == 1.0).all()
    t = reads[4].qfdrp(1, 2, 40, 4)                                           # qfdrp.rs:411-427 (exact)
    assert t.pos[:, 0].tolist() == [0, 2, 4, 6, 13, 15, 17, 19]
    assert (t.val == f32(8.0 / 15.0)).all()
    assert len(reads[5].qfdrp(1, 2, 40, 4)) == 0                              # qfdrp.rs:429-439


def test_qfdrp_hamming_via_pairs(reads):
    """qfdrp.rs:269-306: hamming(read0, read k) == popcount(k) on test1; num_overlap_cpgs(0,1)==4
    (qfdrp.rs:333-356).  Checked through a two-read qFDRP = ham/ncpg."""
    rec = bamio.read_bam(os.path.join(os.path.dirname(__file__), "golden", "test1.bam"))
    for k in range(1, 16):
        r = pyoracle.Reads.decode(rec.subset([0, k]))
        t = r.qfdrp(0, 2, 40, 4)
        assert (t.val == f32(bin(k).count("1")) / f32(4)).all()


# ---------------------------------------------------------------- derived CLI goldens (SURVEY 8c tail)
def test_default_cli_values(reads):
    t = reads[1].pdr()  # -d 10 -p 4 -q 10
    lines = ["chr1\t%d\t%d\t%s\t%d\t%d" % (p, p + 2, pyoracle.format_f32(v), c[0], c[1])
             for p, v, c in zip(t.pos[:, 0], t.val, t.cnt)]
    assert lines == ["chr1\t0\t2\t0.875\t2\t14", "chr1\t2\t4\t0.875\t2\t14",
                     "chr1\t4\t6\t0.875\t2\t14", "chr1\t6\t8\t0.875\t2\t14"]
    assert pyoracle.format_f32(reads[1].lpmd()["lpmd"]) == "0.5"
    assert [pyoracle.format_f32(v) for v in reads[1].mhl().val] == ["0.1625"] * 4
    assert [pyoracle.format_f32(v) for v in reads[1].pm().val] == ["0.9375"]
    assert [pyoracle.format_f32(v) for v in reads[1].me().val] == ["1"]
    assert [pyoracle.format_f32(v) for v in reads[1].fdrp().val] == ["0"] * 4   # 8-bp reads < --min-overlap 35
    assert [pyoracle.format_f32(v) for v in reads[1].qfdrp().val] == ["0"] * 4


def test_format_f32():
    cases = {0.875: "0.875", 8.0 / 15.0: "0.53333336", 1.0: "1", 0.0: "0", 0.1625: "0.1625",
             float("nan"): "NaN", 1e-7: "0.0000001", 123456.0: "123456", 0.1: "0.1", 1.0 / 3.0: "0.33333334",
             16777216.0: "16777216", 1.5e10: "15000000000"}
    for v, s in cases.items():
        assert pyoracle.format_f32(v) == s, (v, pyoracle.format_f32(v), s)
    rng = np.random.default_rng(0)
    for v in rng.random(2000, dtype=np.float32):
        s = pyoracle.format_f32(v)
        assert np.float32(s) == v and s == np.format_float_positional(v, unique=True, trim="-")
