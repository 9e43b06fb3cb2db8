"""Input side of the `tag` parity tests.

The reference's golden for `metheor tag` (tests/tag-cli.rs:60-80) is: tag(test.chr19.noXM.sam, hg38.chr19.fa) ==
test.chr19.XM.sam byte for byte, and it ships its own output of that run (test.chr19.metheor_tag_out.sam, identical).
The 58-MB hg38.chr19.fa is not shipped.  What IS shipped pins the bases the run looked at: every read carries MD:Z
(the reference base under each mismatch; the read base elsewhere), so every covered position of chr19 is known
exactly.  The only positions the run read and the reads do not cover are the two flank bases past a read's end (or
before its start for reverse reads) that decide the context of a C in the read's last two columns; those few are
chosen here so that they satisfy the golden letter (Z/z: next base G; X/x: H then G; H/h: H, H) -- inferred, and
counted separately by the tests.  Everything else in the FASTA is N.

tests/golden/test.chr19.XM.sam is the reference's fixture (copied data); the noXM input is that file with the XM:Z
field removed (verified identical to the reference's tests/test.chr19.noXM.sam when this was written)."""
import os
import re

CIGAR_OPS = "MIDNSHP=X"
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


class SamRead:
    __slots__ = ("line", "name", "flag", "rname", "pos", "cigar", "seq", "md", "xm")


def parse_sam(path):
    """-> header lines, reads (with XM split off), the XM-free text"""
    hdr, reads, stripped = [], [], []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\n")
            if line.startswith("@"):
                hdr.append(line); stripped.append(line)
                continue
            f = line.split("\t")
            r = SamRead()
            r.name, r.flag, r.rname, r.pos = f[0], int(f[1]), f[2], int(f[3]) - 1
            r.cigar = [(int(n) << 4) | CIGAR_OPS.index(op) for n, op in re.findall(r"(\d+)([MIDNSHP=X])", f[5])]
            r.seq = f[9]
            r.md = r.xm = None
            keep = []
            for a in f[11:]:
                if a.startswith("XM:Z:"):
                    r.xm = a[5:]
                    continue
                if a.startswith("MD:Z:"):
                    r.md = a[5:]
                keep.append(a)
            r.line = "\t".join(f[:11] + keep)
            stripped.append(r.line)
            reads.append(r)
    return hdr, reads, "\n".join(stripped) + "\n"


def ref_bases_from_md(r):
    """[(position, base)] of the reference under an all-M read (the fixture has no other CIGAR op)"""
    assert all((c & 15) == 0 for c in r.cigar) and r.md is not None
    out, q = [], 0
    for num, sub in re.findall(r"(\d+)|([A-Z]|\^[A-Z]+)", r.md):
        if num:
            for _ in range(int(num)):
                out.append((r.pos + q, r.seq[q].upper())); q += 1
        else:
            assert not sub.startswith("^")
            out.append((r.pos + q, sub)); q += 1
    assert q == len(r.seq), (q, len(r.seq), r.md)
    return out


def rebuild_contig(reads, length):
    """-> bytearray of the contig (N where nothing is known), number of inferred flank bases,
    number of golden letters whose whole context is MD-derived, number of golden context letters"""
    contig = bytearray(b"N" * length)
    known = {}
    for r in reads:
        for p, b in ref_bases_from_md(r):
            assert known.setdefault(p, b) == b, ("MD tags disagree at", p)
    for p, b in known.items():
        contig[p] = ord(b)
    allowed = {}

    def need(p, bases):
        if p < 0 or p >= length:
            return
        if p in known:
            assert known[p] in bases, ("golden letter contradicts the MD-derived base at", p)
        else:
            allowed[p] = allowed.get(p, set("ACGT")) & set(bases)

    n_letters = n_interior = 0
    for r in reads:
        rev = bool(r.flag & 16)
        d = -1 if rev else 1
        h = "TGA" if rev else "ACT"          # H (not G) as seen from the read's strand, in forward-strand letters
        g = "C" if rev else "G"
        for j, ch in enumerate(r.xm):
            if ch in ".":
                continue
            p = r.pos + j
            n_letters += 1
            if all(0 <= p + d * k < length and (p + d * k) in known for k in (1, 2)):
                n_interior += 1
            if ch in "Zz":
                need(p + d, g)
            elif ch in "Xx":
                need(p + d, h); need(p + 2 * d, g)
            elif ch in "Hh":
                need(p + d, h); need(p + 2 * d, h)
    for p, s in allowed.items():
        assert s, ("no base satisfies the golden letters at", p)
        contig[p] = ord(sorted(s)[0])
    return contig, len(allowed), n_interior, n_letters


def write_fasta(path, name, seq, width=60, with_fai=True):
    with open(path, "wb") as fh:
        fh.write(b">" + name.encode() + b"\n")
        off = fh.tell()
        for i in range(0, len(seq), width):
            fh.write(bytes(seq[i:i + width]) + b"\n")
    if with_fai:
        with open(path + ".fai", "w") as fh:
            fh.write("%s\t%d\t%d\t%d\t%d\n" % (name, len(seq), off, width, width + 1))


def golden(golden_dir):
    hdr, reads, noxm_text = parse_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    ln = [int(x.split("LN:")[1].split("\t")[0]) for x in hdr if x.startswith("@SQ")][0]
    return hdr, reads, noxm_text, ln
