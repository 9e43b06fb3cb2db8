"""PDR / MHL / FDRP / qFDRP on input that is not coordinate-sorted (metheor_amd/csrc/mth_fileorder.hip): randomised files and
parameters through the CLI against the oracle, which streams the records in FILE order as the reference does (pdr.rs:139-178,
mhl.rs:155-173, fdrp.rs:197-223, qfdrp.rs:209-235).  Includes the reservoir branch (max_depth below the depth; oracle and device
share the counter-based draw, METHEOR_SEED), min_depth 0 (NaN rows), spans beyond FDRP's 201-bp window, --cpg-set."""
import os
import subprocess

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metheor_amd", "metheor")


def run(env, *args):
    return subprocess.run([EXE, *[str(a) for a in args]], capture_output=True, text=True, cwd=ROOT, timeout=600, env=dict(os.environ, **env))


def rows(t, names):
    return "".join("%s\t%d\t%d\t%s\n" % (names[ti], p, p + 2, pyoracle.format_f32(v)) for ti, p, v in zip(t.tid, t.pos[:, 0], t.val))


@pytest.mark.parametrize("seed", range(8))
def test_random_unsorted_files(tmp_path, seed):
    from metheor_amd import synth
    rng = np.random.default_rng(7000 + seed)
    read_len = int(rng.choice([40, 100, 150, 230]))
    cs = [synth.make_contig(t, ln, n, float(rng.choice([0.02, 0.05, 0.1])), rng, read_len=read_len)
          for t, (ln, n) in enumerate([(int(rng.integers(3_000, 20_000)), int(rng.integers(300, 2_500))), (int(rng.integers(2_000, 8_000)), int(rng.integers(100, 1_200)))])]
    recs = [util.contig_to_records(c, "ctg%d" % k) for k, c in enumerate(cs)]
    rec = bamio.Records([r.refs[0] for r in recs], np.concatenate([recs[0].tid, recs[1].tid + 1]), np.concatenate([r.pos for r in recs]),
                        np.concatenate([r.flag for r in recs]), np.concatenate([r.mapq for r in recs]), recs[0].cigars + recs[1].cigars, recs[0].xms + recs[1].xms)
    n = len(rec.tid)
    kind = seed % 4
    if kind == 0:
        perm = rng.permutation(n)                                                   # fully shuffled: mostly one-read segments
    elif kind == 1:
        perm = np.argsort(np.arange(n) % int(rng.integers(2, 9)), kind="stable")    # a few interleaved sorted strides
    elif kind == 2:
        perm = np.arange(n)
        for _ in range(int(rng.integers(1, 6))):                                    # a sorted file with a few blocks moved
            a, b = sorted(rng.integers(0, n, size=2))
            perm = np.concatenate([perm[:a], perm[b:], perm[a:b]])
    else:
        perm = np.argsort(rec.pos + rng.integers(-60, 60, size=n), kind="stable")   # nearly sorted, contigs mixed
    sh = bamio.Records(rec.refs, rec.tid[perm], rec.pos[perm], rec.flag[perm], rec.mapq[perm], [rec.cigars[i] for i in perm], [rec.xms[i] for i in perm])
    bam = str(tmp_path / "u.bam")
    bamio.write_bam(bam, sh)
    names = [r[0] for r in rec.refs]
    cpg_set = None
    if seed % 3 == 0:                                                               # --cpg-set: every second called site
        soa = pyoracle.Reads.decode(sh).soa()
        keys = np.unique((soa["tid"].astype(np.int64)[np.repeat(np.arange(len(soa["tid"])), np.diff(soa["cpg_off"].astype(np.int64)))] << 32) | (soa["cpg_pos"] & 0x7fffffff))[::2]
        cpg_set = [(int(k >> 32), int(k & 0xffffffff)) for k in keys]
        bed = tmp_path / "set.bed"
        bed.write_text("".join("%s\t%d\t%d\n" % (names[t], p, p + 2) for t, p in cpg_set))
    reads = pyoracle.Reads.decode(sh, cpg_set=cpg_set)
    extra = ["-c", str(tmp_path / "set.bed")] if cpg_set is not None else []
    o = tmp_path / "o.tsv"
    mq = int(rng.choice([0, 10, 43]))
    for _ in range(2):
        d, p = int(rng.choice([0, 1, 2, 5])), int(rng.choice([0, 1, 2, 4]))
        r = run({}, "pdr", "-i", bam, "-o", o, "-d", d, "-p", p, "-q", mq, *extra)
        assert r.returncode == 0, r.stderr
        assert o.read_text() == util.oracle_tsv_pdr(reads, names, min_depth=d, min_cpgs=p, min_qual=mq), ("pdr", d, p, mq)
        r = run({}, "mhl", "-i", bam, "-o", o, "-d", d, "-p", max(p, 1), "-q", mq, *extra)
        assert r.returncode == 0, r.stderr
        assert o.read_text() == rows(reads.mhl(min_depth=d, min_cpgs=max(p, 1), min_qual=mq), names), ("mhl", d, p, mq)
        D, l, sd = int(rng.choice([2, 8, 40, 200])), int(rng.choice([0, 10, 35])), int(rng.integers(0, 1000))
        for sub in ("fdrp", "qfdrp"):
            r = run({"METHEOR_SEED": str(sd)}, sub, "-i", bam, "-o", o, "-q", mq, "-d", d, "-D", D, "-l", l, *extra)
            try:
                tab = getattr(reads, sub)(min_qual=mq, min_depth=d, max_depth=D, min_overlap=l, seed=sd)
            except pyoracle.ReferencePanic:          # reads beyond 201 bases can hit fdrp.rs:70-72; the engine refuses them the same way
                assert r.returncode == 101 and "fdrp.rs:70-72" in r.stderr, (sub, r.returncode, r.stderr)
                continue
            assert r.returncode == 0, (sub, r.stderr)
            assert o.read_text() == rows(tab, names), (sub, d, D, l, mq, sd)


def _deep_unsorted(tmp_path, n_reads, seed=5):
    from metheor_amd import synth
    rng = np.random.default_rng(seed)
    c = synth.make_contig(0, 900, n_reads, 0.05, rng)                # 2000 reads: ~330 over every position
    r0 = util.contig_to_records(c, "ctg0")
    # nearly sorted (blocks of the sorted file moved): segments stay long, so the stored-read counts do exceed 256
    perm = np.arange(len(r0.tid))
    for _ in range(3):
        a, b = sorted(rng.integers(0, len(perm), size=2))
        perm = np.concatenate([perm[:a], perm[b:], perm[a:b]])
    sh = bamio.Records(r0.refs, r0.tid[perm], r0.pos[perm], r0.flag[perm], r0.mapq[perm], [r0.cigars[i] for i in perm], [r0.xms[i] for i in perm])
    bam = str(tmp_path / "u.bam")
    bamio.write_bam(bam, sh)
    return bam, sh


@pytest.mark.parametrize("D", [300, 1000])
def test_deep_sites_on_unsorted_input(tmp_path, D):
    """--max-depth above 256 on input that is not coordinate-sorted: the stored reads of a site live in a per-wave row of HBM scratch
    (round 5; rounds 3-4 refused it).  D = 300 is below the depth of most sites (reservoir branch, shared counter-based draw),
    D = 1000 above it."""
    bam, sh = _deep_unsorted(tmp_path, 2000)
    reads = pyoracle.Reads.decode(sh)
    o = tmp_path / "o.tsv"
    deepest = 0
    for sub in ("fdrp", "qfdrp"):
        r = run({"METHEOR_SEED": "11"}, sub, "-i", bam, "-o", o, "-q", 10, "-d", 2, "-D", D, "-l", 35)
        assert r.returncode == 0, r.stderr
        tab = getattr(reads, sub)(min_qual=10, min_depth=2, max_depth=D, min_overlap=35, seed=11)
        assert o.read_text() == rows(tab, ["ctg0"]), (sub, D)
        deepest = max(deepest, int(tab.cnt.max()) if tab.cnt.size else 0)
    # the case is only worth its name if some site did hold more reads than the LDS slots
    assert deepest > 256, deepest


def test_limits_are_loud(tmp_path):
    """--max-depth above 16384 (the sorted path's limit too): exit 101 with the reason"""
    bam, _ = _deep_unsorted(tmp_path, 200)
    r = run({}, "fdrp", "-i", bam, "-o", tmp_path / "o.tsv", "-D", 20000)
    assert r.returncode == 101 and "max_depth" in r.stderr.replace("-", "_"), r.stderr


def _panic_records():
    """a reverse-strand read of 210 bases whose first base and base 202 are CpG calls: sites start - 1 and start + 201"""
    xm = bytearray(b"." * 210)
    xm[0] = ord("z"); xm[202] = ord("Z"); xm[100] = ord("z")
    mk = lambda pos, flag, x: (0, pos, flag, 40, np.array([(len(x) << 4) | 0], np.uint32), bytes(x))
    rows_ = [mk(1000, 16, xm), mk(1005, 0, b"z" + b"." * 48 + b"Z"), mk(1300, 0, b"Z" + b"." * 30 + b"z")]
    return bamio.Records([("ctg0", 5000)], np.array([r[0] for r in rows_], np.int32), np.array([r[1] for r in rows_], np.int32),
                         np.array([r[2] for r in rows_], np.uint16), np.array([r[3] for r in rows_], np.uint8), [r[4] for r in rows_], [r[5] for r in rows_])


@pytest.mark.parametrize("order", ["sorted", "unsorted"])
@pytest.mark.parametrize("sub", ["fdrp", "qfdrp"])
def test_reference_window_panic_is_reproduced(tmp_path, order, sub):
    """fdrp.rs:70-72: new_read[-1] -- the reference dies with an index panic; the engine exits 101 naming it, on both paths;
    below the mapq cut (fdrp.rs:205) the read is skipped and the run succeeds"""
    rec = _panic_records()
    if order == "unsorted":
        perm = np.array([2, 0, 1])
        rec = bamio.Records(rec.refs, rec.tid[perm], rec.pos[perm], rec.flag[perm], rec.mapq[perm], [rec.cigars[i] for i in perm], [rec.xms[i] for i in perm])
    bam = str(tmp_path / "p.bam")
    bamio.write_bam(bam, rec)
    with pytest.raises(pyoracle.ReferencePanic):
        getattr(pyoracle.Reads.decode(rec), sub)(min_depth=1)
    r = run({}, sub, "-i", bam, "-o", tmp_path / "o.tsv", "-d", 1)
    assert r.returncode == 101 and "fdrp.rs:70-72" in r.stderr, (r.returncode, r.stderr)
    r = run({}, sub, "-i", bam, "-o", tmp_path / "o.tsv", "-d", 1, "-q", 41)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "o.tsv").read_text() == rows(getattr(pyoracle.Reads.decode(rec), sub)(min_depth=1, min_qual=41), ["ctg0"])
