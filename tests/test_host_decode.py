"""Product host (C++ BGZF/BAM reader + XM decode, libmetheor_host.so) against the oracle's decode
(oracle/bamio.py pure-Python loader + orc_decode): two independently written implementations of
readutil.rs:24-53, 323-345, 87-95, 347-374 must produce the same SoA.  No GPU needed."""
import os
import re

import numpy as np
import pytest

from metheor_amd import hostapi
from oracle import bamio, pyoracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("tid", "start", "end", "mapq", "fwd", "cpg_off", "cpg_pos", "cpg_rel")


def same_soa(a, b):
    for k in KEYS:
        assert a[k].shape == b[k].shape, k
        assert (a[k] == b[k]).all(), k


def test_header_symbols_exported():
    src = open(os.path.join(ROOT, "include", "metheor_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(mth_host_[a-z0-9_]+)\s*\(", src)))
    L = hostapi.lib()
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(hostapi.SYMBOLS) == syms


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_reference_fixtures(golden_dir, k):
    path = os.path.join(golden_dir, "test%d.bam" % k)
    f = hostapi.BamFile(path)
    assert f.refs == [("chr1", 248956422)]
    same_soa(f.decode(), pyoracle.Reads.decode(bamio.read_bam(path)).soa())


def test_real_rrbs_reads_through_bam(golden_dir, tmp_path):
    """the reference's 1000-read RRBS SAM fixture (676 reverse-strand reads), round-tripped to BAM"""
    rec = bamio.read_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    p = str(tmp_path / "rrbs.bam")
    bamio.write_bam(p, rec)
    f = hostapi.BamFile(p)
    assert f.refs == [("chr19", 58617616)]
    got = f.decode()
    same_soa(got, pyoracle.Reads.decode(rec).soa())
    assert (got["fwd"] == 0).sum() == 676 and len(got["tid"]) == 1000


def _weird_records():
    """CIGARs and flags no reference fixture has: indels, clips, ref-skips, =/X, paired flags,
    a short XM, an unaligned record, three contigs; > 64 KiB so that records straddle BGZF blocks"""
    M, I, D, N, S, H, P, EQ, X = range(9)
    rng = np.random.default_rng(99)
    refs = [("chrA", 100000), ("chrB", 50000), ("chrC", 2000)]
    tid, pos, flag, mapq, cigars, xms = [], [], [], [], [], []

    def add(t, p, fl, mq, cig, xm=None):
        qlen = sum(l for op, l in cig if op in (M, I, S, EQ, X))
        if xm is None:
            xm = rng.choice(np.frombuffer(b"zZ.hHxX.", dtype=np.uint8), size=qlen).astype(np.uint8).tobytes()
        tid.append(t); pos.append(p); flag.append(fl); mapq.append(mq)
        cigars.append([(l << 4) | op for op, l in cig]); xms.append(xm)

    add(0, 10, 0, 40, [(M, 20)], b"Z.z.Z.z.Z.z.Z.z.Z.z.")
    add(0, 12, 16, 30, [(S, 3), (M, 10), (I, 2), (M, 8), (D, 4), (M, 6), (S, 2)])
    add(0, 15, 99, 42, [(H, 5), (M, 7), (N, 100), (M, 9), (H, 1)])
    add(0, 15, 147, 42, [(EQ, 6), (X, 1), (EQ, 8)])
    add(0, 20, 83, 9, [(M, 12)])                       # paired reverse: "reverse" rule
    add(0, 20, 163, 60, [(M, 12)])
    add(0, 30, 0, 40, [(M, 30)], b"zzZZ")              # XM shorter than the query
    add(0, 40, 16, 40, [(I, 4), (M, 5)])               # leading insertion
    add(0, 41, 0, 40, [(S, 10)])                       # nothing aligned: start = end = -1
    add(0, 50, 0, 40, [(M, 5), (P, 2), (M, 5)])
    for _ in range(3000):
        t = int(rng.integers(0, 3))
        ln = refs[t][1]
        ops = [(M, int(rng.integers(5, 60)))]
        for _k in range(int(rng.integers(0, 4))):
            ops.append((int(rng.choice([I, D, N, S, EQ, X])), int(rng.integers(1, 9))))
            ops.append((M, int(rng.integers(1, 40))))
        add(t, int(rng.integers(0, ln - 600)), int(rng.choice([0, 16, 99, 147, 83, 163, 1024])), int(rng.integers(0, 61)), ops)
    order = sorted(range(len(tid)), key=lambda i: (tid[i], pos[i]))
    return bamio.Records(refs, [tid[i] for i in order], [pos[i] for i in order], [flag[i] for i in order],
                         [mapq[i] for i in order], [cigars[i] for i in order], [xms[i] for i in order])


def test_cigar_and_strand_rules(tmp_path):
    rec = _weird_records()
    p = str(tmp_path / "weird.bam")
    bamio.write_bam(p, rec)
    assert os.path.getsize(p) > 70000
    got = hostapi.BamFile(p).decode()
    want = pyoracle.Reads.decode(rec).soa()
    same_soa(got, want)
    assert (got["start"] == -1).any() and (got["fwd"] == 0).any() and (got["fwd"] == 1).any()
    # the pure-Python loader reads back what it wrote (guards the writer itself)
    same_soa(pyoracle.Reads.decode(bamio.read_bam(p)).soa(), want)


def test_cpg_set_filter(tmp_path):
    rec = _weird_records()
    p = str(tmp_path / "weird.bam")
    bamio.write_bam(p, rec)
    full = pyoracle.Reads.decode(rec).soa()
    pos = full["cpg_pos"] & 0x7fffffff
    tid_of_call = np.repeat(full["tid"], np.diff(full["cpg_off"]).astype(np.int64))
    rng = np.random.default_rng(3)
    pick = rng.random(len(pos)) < 0.3
    sites = sorted(set(zip(tid_of_call[pick].tolist(), pos[pick].tolist())))
    bed = str(tmp_path / "set.bed")
    with open(bed, "w") as fh:
        for t, q in sites:
            fh.write("%s\t%d\t%d\n" % (rec.refs[t][0], q, q + 2))
    got = hostapi.BamFile(p).decode(cpg_set=bed)
    want = pyoracle.Reads.decode(rec, cpg_set=sites).soa()
    same_soa(got, want)
    assert 0 < len(got["cpg_pos"]) < len(full["cpg_pos"])
    # relpos is preserved by the filter (readutil.rs:87-95)
    assert got["cpg_rel"].max() > 10
    # an empty set filters every call
    empty = str(tmp_path / "empty.bed")
    open(empty, "w").close()
    assert len(hostapi.BamFile(p).decode(cpg_set=empty)["cpg_pos"]) == 0


def test_error_behaviour(golden_dir, tmp_path):
    # bamutil.rs:7-9 + tests/pdr-cli.rs:26-31
    with pytest.raises(hostapi.HostError) as e:
        hostapi.BamFile("tests/no_such.bam")
    assert "Error opening BAM file" in str(e.value) and "file not found" in str(e.value) and "no_such.bam" in str(e.value)
    # tests/cli_error_handling.rs:214-227 (a text file) and :327-347 (zero bytes)
    for content in (b"[package]\nname = \"metheor\"\n", b""):
        q = tmp_path / "x.bam"
        q.write_bytes(content)
        with pytest.raises(hostapi.HostError) as e:
            hostapi.BamFile(str(q))
        assert "Error opening BAM file" in str(e.value)
    # readutil.rs:46,50
    rec = bamio.read_bam(os.path.join(golden_dir, "test1.bam"))
    rec.xms[3] = None
    q = str(tmp_path / "noxm.bam")
    bamio.write_bam(q, rec)
    with pytest.raises(hostapi.HostError) as e:
        hostapi.BamFile(q).decode()
    assert "Error reading XM tag in BAM record" in str(e.value)
    with pytest.raises(RuntimeError):
        pyoracle.Reads.decode(rec)
    # readutil.rs:356 and the unwrap() on an unknown contig (bamutil.rs:24)
    f = hostapi.BamFile(os.path.join(golden_dir, "test1.bam"))
    with pytest.raises(hostapi.HostError) as e:
        f.decode(cpg_set=str(tmp_path / "nonexistent.bed"))
    assert "Could not read target CpG file" in str(e.value)
    bad = tmp_path / "bad.bed"
    bad.write_text("chrZZ\t5\t7\n")
    with pytest.raises(hostapi.HostError):
        hostapi.BamFile(os.path.join(golden_dir, "test1.bam")).decode(cpg_set=str(bad))
    # a truncated file is an error, not a short result
    data = open(os.path.join(golden_dir, "test4.bam"), "rb").read()
    t = tmp_path / "trunc.bam"
    t.write_bytes(data[:200])
    with pytest.raises(hostapi.HostError):
        hostapi.BamFile(str(t)).decode()


def test_format_f32_three_ways():
    """product (std::to_chars) vs oracle (shortest %.Ne search) vs numpy: Rust `{}` of an f32"""
    cases = {0.875: "0.875", 8.0 / 15.0: "0.53333336", 1.0: "1", 0.0: "0", 0.1625: "0.1625",
             float("nan"): "NaN", 1e-7: "0.0000001", 0.9375: "0.9375", 1.0 / 3.0: "0.33333334", 16777216.0: "16777216"}
    for v, s in cases.items():
        assert hostapi.format_f32(v) == s, (v, hostapi.format_f32(v))
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.random(3000, dtype=np.float32), (rng.random(500, dtype=np.float32) * 1e-4).astype(np.float32),
                           (rng.integers(0, 1000, 500) / rng.integers(1, 1000, 500)).astype(np.float32)])
    for v in vals:
        a = hostapi.format_f32(v)
        assert a == pyoracle.format_f32(v) == np.format_float_positional(v, unique=True, trim="-"), v
        assert np.float32(a) == v


def test_thread_count_invariance(tmp_path, monkeypatch):
    """the multi-threaded decoder returns the same SoA for any thread count (records straddle
    BGZF blocks and decode windows; realistic sequence/quality bytes)"""
    rec = _weird_records()
    rec2 = bamio.Records(rec.refs, np.tile(rec.tid, 6), np.tile(rec.pos, 6), np.tile(rec.flag, 6), np.tile(rec.mapq, 6),
                         rec.cigars * 6, rec.xms * 6)
    p = str(tmp_path / "big.bam")
    bamio.write_bam(p, rec2, realistic=True)
    assert os.path.getsize(p) > 1_000_000
    want = pyoracle.Reads.decode(rec2).soa()
    monkeypatch.setenv("METHEOR_DECODE_WINDOW_MB", "1")     # several decode windows: records are carried across them
    for nt in ("1", "2", "7", "64"):
        monkeypatch.setenv("METHEOR_THREADS", nt)
        same_soa(hostapi.BamFile(p).decode(), want)
    monkeypatch.delenv("METHEOR_DECODE_WINDOW_MB")
    same_soa(hostapi.BamFile(p).decode(), want)


def test_fast_synthetic_writer_round_trip(tmp_path):
    """the C++ synthetic-BAM writer (bench tooling): what it writes decodes back to the generator's SoA,
    through the product reader and through the independent pure-Python loader"""
    from metheor_amd import synth
    c = synth.make_contig(0, 300_000, 40_000, 0.03, np.random.default_rng(9))
    p = str(tmp_path / "fast.bam")
    hostapi.write_synthetic_bam(p, c, contig="chrT", seed=3, threads=5)
    want = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c)).soa()
    f = hostapi.BamFile(p)
    assert f.refs == [("chrT", 300_000)]
    same_soa(f.decode(), want)
    same_soa(pyoracle.Reads.decode(bamio.read_bam(p)).soa(), want)
    assert os.path.getsize(p) > 40_000 * 100        # realistic compression (> 100 B/read)


def random_aux_fields(rng, n):
    """n (before, after) byte strings of random BAM aux fields of every type -- A c C s S i I f Z H and B arrays of every
    subtype -- none of them tagged XM (SAM spec 4.2.4); the decoders must step over them to find XM:Z"""
    import struct

    def field():
        tag = bytes(rng.choice(np.frombuffer(b"ABCDEFGHIJKLMNOPQRSTUVWYZ", np.uint8), size=2))     # no 'X' first letter: never XM
        t = rng.choice(list("AcCsSiIfZHB"))
        if t == "A":
            return tag + b"A" + bytes([int(rng.integers(33, 127))])
        if t in "cC":
            return tag + t.encode() + bytes([int(rng.integers(0, 256))])
        if t in "sS":
            return tag + t.encode() + struct.pack("<H", int(rng.integers(0, 65536)))
        if t in "iI":
            return tag + t.encode() + struct.pack("<I", int(rng.integers(0, 2**32)))
        if t == "f":
            return tag + b"f" + struct.pack("<f", float(rng.normal()))
        if t == "Z":      # strings that look like XM strings, of lengths around the 4-byte scan granularity
            return tag + b"Z" + bytes(rng.choice(np.frombuffer(b"zZ.xhXHMZ:", np.uint8), size=int(rng.integers(0, 40)))) + b"\0"
        if t == "H":
            return tag + b"H" + b"".join(b"%02X" % int(v) for v in rng.integers(0, 256, size=int(rng.integers(0, 9)))) + b"\0"
        sub = rng.choice(list("cCsSiIf"))
        cnt = int(rng.integers(0, 12))
        w = {"c": 1, "C": 1, "s": 2, "S": 2, "i": 4, "I": 4, "f": 4}[sub]
        return tag + b"B" + sub.encode() + struct.pack("<i", cnt) + bytes(rng.integers(0, 256, size=cnt * w, dtype=np.uint8))

    return [(b"".join(field() for _ in range(int(rng.integers(0, 5)))), b"".join(field() for _ in range(int(rng.integers(0, 4)))))
            for _ in range(n)]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_aux_fields(tmp_path, seed):
    rec = _weird_records()
    rec.aux_extra = random_aux_fields(np.random.default_rng(500 + seed), len(rec))
    p = str(tmp_path / "aux.bam")
    bamio.write_bam(p, rec)
    want = pyoracle.Reads.decode(rec).soa()
    same_soa(hostapi.BamFile(p).decode(), want)
    same_soa(pyoracle.Reads.decode(bamio.read_bam(p)).soa(), want)      # (the oracle's own loader steps over them too)


def crafted_aux_cases():
    """aux blocks a hostile file can carry in front of XM:Z: a B-array count whose byte length wraps a 32-bit
    cursor back onto the same tag (5 + 0x3FFFFFFE*4 == -3 mod 2^32: a scanner doing the sum in 32 bits never leaves the
    record), counts that merely run past the record, and an unknown B sub-type.  htslib's skip_aux rejects all of them, so
    Record::aux(b"XM") is Err and the reference panics with the XM text (readutil.rs:46-48)."""
    import struct
    return {
        "wrap_to_same_tag": b"XXBi" + struct.pack("<I", 0x3FFFFFFE),
        "wrap_short": b"XXBs" + struct.pack("<I", 0x7FFFFFFE) + b"\0" * 6,
        "past_the_end": b"YYBC" + struct.pack("<I", 5000),
        "huge_float_array": b"YYBf" + struct.pack("<I", 0xFFFFFFFF),
        "unknown_subtype": b"YYBq" + struct.pack("<I", 1) + b"\0\0\0\0",
    }


@pytest.mark.timeout(60)
@pytest.mark.parametrize("case", sorted(crafted_aux_cases()))
def test_crafted_aux_is_rejected_not_looped(golden_dir, tmp_path, case):
    rec = bamio.read_bam(os.path.join(golden_dir, "test1.bam"))
    rec.aux_extra = [(b"NMC\x00", b"XRZCT\0")] * len(rec)
    rec.aux_extra[2] = (crafted_aux_cases()[case], b"")
    p = str(tmp_path / "crafted.bam")
    bamio.write_bam(p, rec)
    with pytest.raises(hostapi.HostError) as e:
        hostapi.BamFile(p).decode()
    assert "Error reading XM tag in BAM record" in str(e.value)


def test_multi_contig_synthetic_writer_round_trip(tmp_path):
    """the multi-contig form of the C++ synthetic-BAM writer (config 3 / 5 test input): decodes back to the generator's SoA"""
    from metheor_amd import synth
    rng = np.random.default_rng(12)
    names = ["cA", "cEmpty", "cB"]
    cs = [synth.make_contig(0, 400_000, 9_000, 0.02, rng), synth.make_contig(2, 150_000, 4_000, 0.03, rng)]
    p = str(tmp_path / "multi.bam")
    hostapi.write_synthetic_bam_multi(p, cs, names, seed=5, threads=3)
    want = pyoracle.Reads.from_soa(*synth.concat_oracle_soa(cs)).soa()
    f = hostapi.BamFile(p)
    assert f.refs == [("cA", 400_000), ("cEmpty", 1000), ("cB", 150_000)]
    same_soa(f.decode(), want)
    same_soa(pyoracle.Reads.decode(bamio.read_bam(p)).soa(), want)


def hand_worked_records():
    """Records whose decode is worked out BY HAND from the SAM spec (CIGAR ops: M/=/X consume query and reference, I/S query
    only, D/N reference only, H/P nothing) and readutil.rs:24-33, 323-345 (start/end = first/last aligned reference position;
    a z/Z at query offset q with an aligned reference position p is a call at p for flags in {0, 99, 147}, at p - 1 otherwise;
    relpos = q; insertions and soft clips have no reference position and are skipped) -- literal expectations, independent of
    every decoder in this repository.  -> (Records, expected) with expected = per record (start, end, fwd, [(pos, meth, rel)])"""
    M, I, D, N, S, H, P, EQ, X = range(9)
    c = lambda *ops: [(l << 4) | op for op, l in ops]
    recs = [
        # 2S 3M 2I 2M 3D 2M 1S at 100: query 0-1 clipped, 2-4 -> 100-102, 5-6 inserted, 7-8 -> 103-104, (105-107 deleted), 9-10 -> 108-109, 11 clipped
        (0, 100, 0, 40, c((S, 2), (M, 3), (I, 2), (M, 2), (D, 3), (M, 2), (S, 1)), b"Z.z.ZZ.xZz.z", (100, 109, 1, [(100, 0, 2), (102, 1, 4), (104, 1, 8), (108, 0, 9)])),
        # the same alignment on the reverse strand (flag 16): every call one base to the left, start / end unchanged
        (0, 100, 16, 40, c((S, 2), (M, 3), (I, 2), (M, 2), (D, 3), (M, 2), (S, 1)), b"Z.z.ZZ.xZz.z", (100, 109, 0, [(99, 0, 2), (101, 1, 4), (103, 1, 8), (107, 0, 9)])),
        # 5H 4M 100N 4M 2H at 200, flag 99 (forward rule): hard clips consume nothing, the ref-skip jumps 204..303
        (0, 200, 99, 30, c((H, 5), (M, 4), (N, 100), (M, 4), (H, 2)), b"Z..zZ..z", (200, 307, 1, [(200, 1, 0), (203, 0, 3), (304, 1, 4), (307, 0, 7)])),
        # 3= 1X 2= at 50, flag 163 (not one of 0 / 99 / 147: reverse rule)
        (0, 50, 163, 20, c((EQ, 3), (X, 1), (EQ, 2)), b".Z.z..", (50, 55, 0, [(50, 1, 1), (52, 0, 3)])),
        # flag 147 is a forward-rule flag; a leading insertion and a padding op: 2I 3M 1P 2M at 10 -> query 2-4 -> 10-12, 5-6 -> 13-14
        (0, 10, 147, 5, c((I, 2), (M, 3), (P, 1), (M, 2)), b"zZz..ZZ", (10, 14, 1, [(10, 0, 2), (13, 1, 5), (14, 1, 6)])),
        # flag 83 (reverse rule) with the first call at the read's first base: position start - 1
        (0, 1000, 83, 60, c((M, 4),), b"Z..z", (1000, 1003, 0, [(999, 1, 0), (1002, 0, 3)])),
        # XM shorter than the query (only the first 3 offsets exist), other context letters are not calls
        (0, 2000, 0, 60, c((M, 8),), b"hZx", (2000, 2007, 1, [(2001, 1, 1)])),
    ]
    rec = bamio.Records([("chrH", 10_000)], [r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs],
                        [r[4] for r in recs], [r[5] for r in recs])
    return rec, [r[6] for r in recs]


def check_hand_worked(soa, expected):
    off = soa["cpg_off"].astype(np.int64)
    for i, (st, en, fwd, calls) in enumerate(expected):
        assert (int(soa["start"][i]), int(soa["end"][i]), int(soa["fwd"][i])) == (st, en, fwd), i
        got = [(int(p & 0x7fffffff), int(p >> 31), int(r)) for p, r in zip(soa["cpg_pos"][off[i]:off[i + 1]], soa["cpg_rel"][off[i]:off[i + 1]])]
        assert got == calls, (i, got, calls)


def test_hand_worked_cigar_and_strand_rules(tmp_path):
    rec, expected = hand_worked_records()
    check_hand_worked(pyoracle.Reads.decode(rec).soa(), expected)              # the oracle's decoder
    p = str(tmp_path / "hand.bam")
    bamio.write_bam(p, rec)
    check_hand_worked(hostapi.BamFile(p).decode(), expected)                    # the product's host decoder
    check_hand_worked(pyoracle.Reads.decode(bamio.read_bam(p)).soa(), expected) # ... and through the Python BAM loader
