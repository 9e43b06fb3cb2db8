"""loader for tests/golden/unpinned_cases.json.gz (expected values from tools/gen_golden_unpinned.py, the Python transliteration
written from the reference's Rust) -> oracle.bamio.Records"""
import gzip
import json
import os

import numpy as np

from oracle import bamio

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def load():
    with gzip.open(os.path.join(HERE, "golden", "unpinned_cases.json.gz"), "rb") as f:
        return json.loads(f.read().decode())


def refs_of(case):
    n = max(r["tid"] for r in case["records"]) + 1
    return [("ctg%d" % t, 1_000_000) for t in range(n)]


def records_of(case):
    rs = case["records"]
    cig = []
    for r in rs:
        ops, num = [], ""
        for ch in r["cigar"]:
            if ch.isdigit():
                num += ch
            else:
                ops.append((int(num) << 4) | bamio.CIGAR_OPS.index(ch))
                num = ""
        cig.append(ops)
    return bamio.Records(refs_of(case), [r["tid"] for r in rs], [r["pos"] for r in rs], [r["flag"] for r in rs], [r["mapq"] for r in rs],
                         cig, [r["xm"].encode() for r in rs])


def from_bits(b):
    return np.array(b, dtype=np.uint32).view(np.float32)


def same_f32(a, b, tol=None):
    """bit-equal (NaN == NaN), or within tol where the reference itself fixes no summation order (MHL) / calls libm (ME)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    if a.shape != b.shape:
        return False
    eq = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    if tol is not None:
        with np.errstate(invalid="ignore"):
            eq |= np.abs(a.astype(np.float64) - b.astype(np.float64)) <= tol
    return bool(eq.all())
