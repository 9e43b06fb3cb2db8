"""`metheor tag` on the MI355X: the device kernel (mth_tag_records) and the CLI against the reference's
golden output for its 1000-read fixture (tests/tag-cli.rs:60-80) and against the oracle (oracle/tag_oracle.cpp) on
generated records with every CIGAR operation, both strands, paired flags, contig edges and a mixed-case genome."""
import os
import subprocess

import numpy as np
import pytest

from metheor_amd import hostapi
from oracle import pyoracle
from tests import tag_util

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
def chr19(golden_dir, tmp_path_factory):
    hdr, reads, noxm_text, ln = tag_util.golden(golden_dir)
    contig, _, _, _ = tag_util.rebuild_contig(reads, ln)
    d = tmp_path_factory.mktemp("tag")
    noxm = str(d / "test.chr19.noXM.sam")
    open(noxm, "w").write(noxm_text)
    fa = str(d / "chr19.rebuilt.fa")
    tag_util.write_fasta(fa, "chr19", contig)
    return dict(reads=reads, ln=ln, contig=bytes(contig), noxm=noxm, fa=fa, want=open(os.path.join(golden_dir, "test.chr19.XM.sam"), "rb").read())


def device_xm(eng, sam_path, contigs, paired=False):
    f = hostapi.BamFile(sam_path)
    eng.tag_set_genome(contigs)
    out = []
    for raw, off in f.windows():
        out += eng.tag_records(raw, off, is_paired_end=paired)
    return out


def test_device_reproduces_the_golden_xm_8864_letters_md_derived_680_via_29_fitted_flank_bases(eng, chr19):
    got = device_xm(eng, chr19["noxm"], [(chr19["ln"], chr19["contig"])])
    assert len(got) == 1000
    bad = [(k, r.xm, g) for k, (r, g) in enumerate(zip(chr19["reads"], got)) if g.decode() != r.xm]
    assert not bad, bad[:3]


def test_cli_tag_golden_bytes_genome_rebuilt_from_md_tags_29_flank_bases_fitted(chr19, tmp_path):
    # tests/tag-cli.rs:60-80: metheor tag -i test.chr19.noXM.sam -o out.sam -g <genome>; out.sam == test.chr19.XM.sam
    out = tmp_path / "test.chr19.metheor_tag_out.sam"
    r = subprocess.run([EXE, "tag", "-i", chr19["noxm"], "-o", str(out), "-g", chr19["fa"]], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "Parsing reference genome...\nDone!\n"        # tag.rs:418, 432
    assert out.read_bytes() == chr19["want"]
    # the same without a .fai next to the FASTA (htslib would build one; here the file is scanned)
    os.remove(chr19["fa"] + ".fai")
    out2 = tmp_path / "again.sam"
    r = subprocess.run([EXE, "tag", "-i", chr19["noxm"], "-o", str(out2), "-g", chr19["fa"]], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and out2.read_bytes() == chr19["want"]


# ---- generated records against the oracle ------------------------------------------------------------------------
OPS = "MIDNSHP=X"


def random_records(rng, contigs, n, paired):
    """-> SAM text, [(tid, pos, flag, packed cigar, seq)]"""
    lines = ["@HD\tVN:1.6\tSO:unsorted"] + ["@SQ\tSN:c%d\tLN:%d" % (t, len(c)) for t, c in enumerate(contigs)]
    recs = []
    for i in range(n):
        tid = int(rng.integers(0, len(contigs)))
        ln = len(contigs[tid])
        nops = int(rng.integers(1, 7))
        ops = []
        for k in range(nops):
            op = str(rng.choice(list("MMMMIDNS=X")))
            if op == "S" and 0 < k < nops - 1:
                op = "M"
            if ops and ops[-1][1] == op:
                continue
            ops.append((int(rng.integers(1, 9 if op != "M" else 40)), op))
        if not any(o in "M=X" for _, o in ops):
            ops.append((int(rng.integers(1, 30)), "M"))
        if rng.random() < 0.1:
            ops = [(3, "H")] + ops
        reflen = sum(l for l, o in ops if o in "MDN=X")
        qlen = sum(l for l, o in ops if o in "MIS=X")
        where = rng.random()
        if where < 0.1:
            pos = int(rng.integers(0, 3))                        # at the contig's first bases (N padding on the left)
        elif where < 0.2:
            pos = max(ln - reflen - int(rng.integers(0, 3)), 0)  # ending at / next to the contig's last base
        else:
            pos = int(rng.integers(0, max(ln - reflen, 1)))
        if pos + reflen > ln:
            pos = max(ln - reflen, 0)
        if paired:
            flag = 1 | int(rng.choice([64, 128])) | (16 if rng.random() < 0.5 else 0) | (32 if rng.random() < 0.5 else 0)
        else:
            flag = 16 if rng.random() < 0.5 else 0
        alphabet = list("ACGT") * 6 + list("N") + (list("RYKM") if rng.random() < 0.05 else [])
        seq = "".join(rng.choice(alphabet, size=qlen))
        cigar = "".join("%d%s" % (l, o) for l, o in ops)
        lines.append("r%d\t%d\tc%d\t%d\t30\t%s\t*\t0\t0\t%s\t*" % (i, flag, tid, pos + 1, cigar, seq))
        recs.append((tid, pos, flag, [(l << 4) | OPS.index(o) for l, o in ops], seq.encode()))
    return "\n".join(lines) + "\n", recs


@pytest.mark.parametrize("seed,paired", [(1, False), (2, False), (3, True), (4, True)])
def test_generated_records_match_the_oracle(eng, tmp_path, seed, paired):
    rng = np.random.default_rng(seed)
    contigs = []
    for ln in (400, 1500, 37):
        c = rng.choice(list(b"ACGT") * 8 + list(b"Nacgt"), size=ln).astype(np.uint8)
        contigs.append(bytes(c))
    text, recs = random_records(rng, contigs, 1500, paired)
    want, keep_lines = [], text.splitlines()[:4]
    body = text.splitlines()[4:]
    for line, (tid, pos, flag, cig, seq) in zip(body, recs):
        w = pyoracle.tag_xm(pos, flag, cig, seq, contigs[tid], is_paired_end=paired)
        if w is None:
            continue                                              # the reference panics there: covered by test_panics
        want.append(w); keep_lines.append(line)
    assert len(want) > 1300
    p = tmp_path / "gen.sam"
    p.write_text("\n".join(keep_lines) + "\n")
    # is_paired_end is the FIRST record's flag (bamutil.rs:27-37); the generator sets 0x1 on every record or on none
    got = device_xm(eng, str(p), [(len(c), c) for c in contigs], paired=paired)
    assert len(got) == len(want)
    bad = [(k, keep_lines[4 + k], w, g) for k, (w, g) in enumerate(zip(want, got)) if w != g]
    assert not bad, bad[:3]
    assert any(b"U" in w or b"u" in w for w in want) and any(b"X" in w for w in want) and any(b"h" in w or b"H" in w for w in want)


@pytest.mark.parametrize("line,why", [
    ("r\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\t*", "an unplaced record: tid2size[&tid] panics (tag.rs:155)"),
    ("r\t0\tc0\t398\t30\t10M\t*\t0\t0\tACGTACGTAC\t*", "the alignment ends more than two bases past the contig (tag.rs:170)"),
    ("r\t16\tc0\t10\t30\t2M1I2M\t*\t0\t0\tAC=GT\t*", "'=' has no complement (tag.rs:24)"),
])
def test_panics_are_errors_not_output(eng, tmp_path, line, why):
    import metheor_amd
    contig = b"ACGT" * 100
    p = tmp_path / "p.sam"
    p.write_text("@HD\tVN:1.6\n@SQ\tSN:c0\tLN:400\n" + line + "\n")
    with pytest.raises(metheor_amd.MthError) as e:
        device_xm(eng, str(p), [(400, contig)])
    assert "tag" in str(e.value), why
    eng.reset()
    # and the CLI leaves with the panic status
    fa = str(tmp_path / "g.fa")
    tag_util.write_fasta(fa, "c0", contig)
    r = subprocess.run([EXE, "tag", "-i", str(p), "-o", str(tmp_path / "o.sam"), "-g", fa], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 101, (r.returncode, r.stderr)


def test_cli_tag_fasta_contig_missing(tmp_path):
    # a contig of the header that the FASTA lacks: expect("Error fetching reference genome sequence.") (tag.rs:427)
    p = tmp_path / "two.sam"
    p.write_text("@HD\tVN:1.6\n@SQ\tSN:c0\tLN:8\n@SQ\tSN:other\tLN:8\nr\t0\tc0\t1\t30\t4M\t*\t0\t0\tACGT\t*\n")
    fa = str(tmp_path / "g.fa")
    tag_util.write_fasta(fa, "c0", b"ACGTACGT", with_fai=False)
    r = subprocess.run([EXE, "tag", "-i", str(p), "-o", str(tmp_path / "o.sam"), "-g", fa], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 101 and "Error fetching reference genome sequence" in r.stderr


def test_cli_tag_bam_input_equals_sam_input(chr19, tmp_path):
    # the same 1000 records as a BAM file (seq / qual / every aux field carried by the host library's SAM -> BAM
    # conversion, written out here): the tagged SAM is the same bytes
    f = hostapi.BamFile(chr19["noxm"])
    bam = tmp_path / "noxm.bam"
    bam.write_bytes(open(f.staged_path(), "rb").read())
    out = tmp_path / "from_bam.sam"
    r = subprocess.run([EXE, "tag", "-i", str(bam), "-o", str(out), "-g", chr19["fa"]], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == chr19["want"]


# ---- a read shorter than its CIGAR: chars().skip().take() comes up short in the reference, it does not panic ------
def _bam_record(tid, pos, flag, cigar, seq, name=b"r"):
    """one BAM record (block_size included) with an arbitrary l_seq, whatever the CIGAR says"""
    import struct
    nib = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}
    packed = bytearray((len(seq) + 1) // 2)
    for k, ch in enumerate(seq.decode()):
        packed[k >> 1] |= nib[ch] << (0 if k & 1 else 4)
    body = struct.pack("<iiBBHHHIiii", tid, pos, len(name) + 1, 30, 4680, len(cigar), flag, len(seq), -1, -1, 0)
    body += name + b"\0" + b"".join(struct.pack("<I", c) for c in cigar) + bytes(packed) + b"\xff" * len(seq)
    return struct.pack("<i", len(body)) + body


def test_seq_star_beside_a_cigar_gives_an_empty_xm(eng, tmp_path):
    # bwa mem -a / bwa-meth secondary alignments: SEQ '*' with a CIGAR.  tag.rs:190-216 take() nothing, every read column is '-'
    contig = b"ACGTCGCGAACCGGTTACGT" * 20
    p = tmp_path / "s.sam"
    p.write_text("@HD\tVN:1.6\n@SQ\tSN:c0\tLN:400\n"
                 "a\t0\tc0\t10\t30\t20M\t*\t0\t0\tCGCGAACCGGTTACGTACGT\t*\n"
                 "b\t256\tc0\t10\t30\t20M\t*\t0\t0\t*\t*\n"
                 "c\t272\tc0\t30\t30\t5M2D10M1I4M\t*\t0\t0\t*\t*\n")
    got = device_xm(eng, str(p), [(400, contig)])
    want = [pyoracle.tag_xm(9, 0, [(20 << 4)], b"CGCGAACCGGTTACGTACGT", contig),
            pyoracle.tag_xm(9, 256, [(20 << 4)], b"", contig),
            pyoracle.tag_xm(29, 272, [(5 << 4), (2 << 4) | 2, (10 << 4), (1 << 4) | 1, (4 << 4)], b"", contig)]
    assert want[1] == b"" and want[2] == b"" and len(want[0]) == 20
    assert got == want
    fa = str(tmp_path / "g.fa")
    tag_util.write_fasta(fa, "c0", contig)
    out = tmp_path / "o.sam"
    r = subprocess.run([EXE, "tag", "-i", str(p), "-o", str(out), "-g", fa], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = out.read_text().splitlines()[2:]
    assert lines[1].endswith("\tXM:Z:") and lines[2].endswith("\tXM:Z:") and lines[0].endswith("XM:Z:" + want[0].decode())


@pytest.mark.parametrize("seed", [11, 12])
def test_reads_shorter_than_their_cigar_match_the_oracle(eng, seed):
    """BAM records whose l_seq is smaller than the CIGAR's M + I (only a BAM can say that; SAM text is refused by htslib and by
    sam_text.cpp): the device walks the read and reference column strings independently, as tag.rs does, and agrees with the
    oracle letter for letter -- including the records on which the reference panics"""
    import metheor_amd
    rng = np.random.default_rng(seed)
    contig = bytes(rng.choice(list(b"ACGT") * 8 + list(b"N"), size=600).astype(np.uint8))
    eng.tag_set_genome([(600, contig)])
    n_ok = n_panic = 0
    for i in range(300):
        ops = []
        for k in range(int(rng.integers(1, 6))):
            op = int(rng.choice([0, 0, 0, 1, 2]))
            if ops and (ops[-1] & 15) == op:
                continue
            ops.append((int(rng.integers(1, 25)) << 4) | op)
        if not any((c & 15) == 0 for c in ops):
            ops.append((int(rng.integers(1, 25)) << 4))
        need = sum(c >> 4 for c in ops if (c & 15) in (0, 1))
        reflen = sum(c >> 4 for c in ops if (c & 15) in (0, 2))
        l_seq = int(rng.integers(0, need + 1))                    # 0 ... exactly enough
        seq = bytes(rng.choice(list(b"ACGTN"), size=l_seq).astype(np.uint8))
        pos = int(rng.integers(0, 600 - reflen))
        flag = 16 if rng.random() < 0.5 else 0
        want = pyoracle.tag_xm(pos, flag, ops, seq, contig)
        raw = _bam_record(0, pos, flag, ops, seq)
        if want is None:
            with pytest.raises(metheor_amd.MthError):
                eng.tag_records(raw, np.array([0, len(raw)], np.uint64))
            eng.reset()
            eng.tag_set_genome([(600, contig)])
            n_panic += 1
        else:
            got = eng.tag_records(raw, np.array([0, len(raw)], np.uint64))
            assert got == [want], (i, ops, l_seq, flag, want, got)
            n_ok += 1
    assert n_ok > 150
