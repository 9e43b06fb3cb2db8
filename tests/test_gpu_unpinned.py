"""The DEVICE path against the second, independent pin (tests/golden/unpinned_cases.json.gz: outputs of tools/gen_golden_unpinned.py,
a Python transliteration written from the reference's Rust sources): each case's records are written as a BAM, run through the
drop-in `metheor` executable (device inflate / record decode / measure kernels / TSV writer) and the TSV is compared with the
expected rows -- integers and PDR / LPMD / PM / FDRP / qFDRP floats bit for bit (Rust `{}` prints the shortest round-trip
decimal, so parsing it back is exact), MHL and ME within 1e-6."""
import os
import subprocess

import numpy as np
import pytest

from oracle import bamio
from tests import unpinned_util as U

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "metheor_amd", "metheor")
CASES = U.load()["cases"]


def run(*args):
    return subprocess.run([EXE, *[str(a) for a in args]], capture_output=True, text=True, cwd=ROOT, timeout=600)


def table(path, ncol):
    rows = [l.split("\t") for l in open(path).read().splitlines()]
    assert all(len(r) == ncol for r in rows), rows[:3]
    return rows


def fl(xs):
    return np.array([float(x) for x in xs], dtype=np.float64).astype(np.float32)      # "NaN" parses; f64 -> f32 is exact for a shortest f32 decimal


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_device_equals_the_transliteration(case, tmp_path):
    rec = U.records_of(case)
    names = [n for n, _ in rec.refs]
    bam = str(tmp_path / "in.bam")
    bamio.write_bam(bam, rec)
    o = str(tmp_path / "o.tsv")
    ex = case["expect"]

    def ok(r, what):
        assert r.returncode == 0, (what, r.stderr)       # (the unsorted sets go through the file-order replay, mth_fileorder.hip)

    for e in ex.get("pdr", []):
        p = e["params"]
        ok(run("pdr", "-i", bam, "-o", o, "-d", p["min_depth"], "-p", p["min_cpgs"], "-q", p["min_qual"]), "pdr")
        t = table(o, 6)
        assert [[names.index(r[0]), int(r[1]), int(r[4]), int(r[5])] for r in t] == [[r[0], r[1], r[3], r[4]] for r in e["rows"]], p
        assert all(int(r[2]) == int(r[1]) + 2 for r in t)
        assert U.same_f32(fl([r[3] for r in t]), U.from_bits([r[2] for r in e["rows"]]))
    for e in ex.get("lpmd", []):
        p = e["params"]
        pf = str(tmp_path / "pairs.tsv")
        ok(run("lpmd", "-i", bam, "-o", o, "-m", p["min_distance"], "-M", p["max_distance"], "-q", p["min_qual"], "-p", pf), "lpmd")
        lines = open(o).read().splitlines()
        assert lines[0] == "name\tlpmd" and lines[1].split("\t")[0] == bam
        assert U.same_f32(fl([lines[1].split("\t")[1]]), U.from_bits([e["lpmd_bits"]]))
        pl = open(pf).read().splitlines()
        assert pl[0] == "chrom\tcpg1\tcpg2\tlpmd\tn_concordant\tn_discordant"
        t = [l.split("\t") for l in pl[1:]]
        assert [[names.index(r[0]), int(r[1]), int(r[2]), int(r[4]), int(r[5])] for r in t] == [[r[0], r[1], r[2], r[4], r[5]] for r in e["pairs"]]
        assert U.same_f32(fl([r[3] for r in t]), U.from_bits([r[3] for r in e["pairs"]]))
        # the table's column sums are the global counters (lpmd.rs:186-194)
        assert [sum(int(r[4]) for r in t), sum(int(r[5]) for r in t)] == e["counts"][:2]
    for e in ex.get("mhl", []):
        p = e["params"]
        ok(run("mhl", "-i", bam, "-o", o, "-d", p["min_depth"], "-p", p["min_cpgs"], "-q", p["min_qual"]), "mhl")
        t = table(o, 4)
        assert [[names.index(r[0]), int(r[1])] for r in t] == [r[:2] for r in e["rows"]], p
        assert U.same_f32(fl([r[3] for r in t]), U.from_bits([r[2] for r in e["rows"]]), tol=1e-6)
    for e in ex.get("quartet", []):
        p = e["params"]
        for sub, col, tol in (("me", 6, 1e-6), ("pm", 7, None)):
            ok(run(sub, "-i", bam, "-o", o, "-d", 0, "-q", p["min_qual"]), sub)
            t = sorted(table(o, 6), key=lambda r: (names.index(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4])))      # unsorted in the reference too (HashMap)
            assert [[names.index(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4])] for r in t] == [r[:5] for r in e["rows"]]
            assert U.same_f32(fl([r[5] for r in t]), U.from_bits([r[col] for r in e["rows"]]), tol=tol)
        # the depth filter of the writer (me.rs:82): rows with >= 3 reads
        ok(run("pm", "-i", bam, "-o", o, "-d", 3, "-q", p["min_qual"]), "pm")
        assert len(open(o).read().splitlines()) == sum(1 for r in e["rows"] if sum(r[5]) >= 3)
    for e in ex.get("fdrp", []):
        p = e["params"]
        for sub, col in (("fdrp", 2), ("qfdrp", 3)):
            ok(run(sub, "-i", bam, "-o", o, "-q", p["min_qual"], "-d", p["min_depth"], "-D", p["max_depth"], "-l", p["min_overlap"]), sub)
            t = table(o, 4)
            assert [[names.index(r[0]), int(r[1])] for r in t] == [r[:2] for r in e["rows"]], (sub, p)
            assert U.same_f32(fl([r[3] for r in t]), U.from_bits([r[col] for r in e["rows"]])), (sub, p)
