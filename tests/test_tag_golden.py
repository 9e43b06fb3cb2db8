"""Pins the `tag` oracle (oracle/tag_oracle.cpp, a restatement of src/tag.rs:130-384) on the reference's own golden
(tests/tag-cli.rs:60-80: 1000 chr19 reads, expected output test.chr19.XM.sam == the shipped test.chr19.metheor_tag_out.sam)
and on hand-worked records whose expected strings are written out here from tag.rs, not computed."""
import os

import pytest

from oracle import pyoracle
from tests import tag_util


@pytest.fixture(scope="module")
def chr19(golden_dir):
    hdr, reads, noxm_text, ln = tag_util.golden(golden_dir)
    contig, n_inferred, n_interior, n_letters = tag_util.rebuild_contig(reads, ln)
    return dict(hdr=hdr, reads=reads, noxm=noxm_text, ln=ln, contig=contig, n_inferred=n_inferred,
                n_interior=n_interior, n_letters=n_letters)


def test_fixture_is_what_the_reference_ships(chr19):
    # 1000 reads, single-end (flags 0 / 16), all-M CIGARs; the noXM text is the XM file minus its XM:Z field
    assert len(chr19["reads"]) == 1000
    assert {r.flag for r in chr19["reads"]} == {0, 16}
    ref_noxm = "/root/reference/tests/test.chr19.noXM.sam"
    if os.path.exists(ref_noxm):                         # only in the build container
        assert open(ref_noxm).read() == chr19["noxm"]
    # 8864 of the 9544 context letters are decided by MD-derived bases alone; the others sit in a read's last two
    # columns and look at one of the 29 inferred flank bases
    assert (chr19["n_interior"], chr19["n_letters"], chr19["n_inferred"]) == (8864, 9544, 29)


def test_oracle_reproduces_the_golden_xm_8864_letters_md_derived_680_via_29_fitted_flank_bases(chr19):
    """What this pins and what it does not: hg38.chr19.fa is not shipped, the contig is rebuilt from the reads' MD:Z
    tags.  8864 of the golden's 9544 context letters look only at MD-derived bases: those are an independent check.  The other 680
    sit in a read's last two columns and look at one of 29 flank bases no read covers, which tests/tag_util.py CHOSE so that the
    golden letter comes out: for them the test is circular and only shows that one consistent choice exists."""
    contig = bytes(chr19["contig"])
    bad = []
    for k, r in enumerate(chr19["reads"]):
        got = pyoracle.tag_xm(r.pos, r.flag, r.cigar, r.seq.encode(), contig, is_paired_end=False)
        if got is None or got.decode() != r.xm:
            bad.append((k, r.name, r.xm, got))
    assert not bad, bad[:3]


# ---- hand-worked records (expected strings derived by hand from tag.rs; M = 0, I = 1, D = 2, N = 3, S = 4) ----------
def cg(*ops):
    return [(n << 4) | "MIDNS".index(o) for n, o in ops]

#             0123456789012345
CONTIG = b"ACGTCAGTCTTACCGAA"


@pytest.mark.parametrize("name,pos,flag,cigar,seq,paired,want", [
    # forward, all matches: C1 in CG -> Z, C4 in CAG -> X, C8 in CTT -> H (tag.rs:343-374); other columns '.'
    ("fwd_contexts", 0, 0, cg((11, "M")), b"ACGTCAGTCTT", False, ".Z..X...H.."),
    # the same read with its Cs converted: T under a reference C -> lower case (tag.rs:349, 358, 367)
    ("fwd_converted", 0, 0, cg((11, "M")), b"ATGTTAGTTTT", False, ".z..x...h.."),
    # a read base that is neither C nor T under a reference C -> '.' (tag.rs:352-354); N in the read -> '.' (tag.rs:268)
    ("fwd_other_base", 0, 0, cg((5, "M")), b"AGGTN", False, "....."),
    # the last two columns look at the two reference bases past the alignment (tag.rs:151-164): read covers 0..3,
    # C at 4 is outside; column 1 (C, next G) -> Z.  Read 8..9: C8 then T9 T10 -> CTT -> H.
    ("fwd_flank", 8, 0, cg((2, "M")), b"CT", False, "H."),
    # reverse strand, single-end: contexts are read on the reverse complement (tag.rs:246-256) and the string is
    # reversed back (tag.rs:386-389).  Forward G2 pairs with C1 (CG) -> Z at column 2; G6: complement context is
    # C then comp(A5)=T, comp(C4)=G -> CTG -> X; G14: comp(C13)=G -> CG -> Z
    ("rev_contexts", 0, 16, cg((16, "M")), b"ACGTCAGTCTTACCGA", False, "..Z...X.......Z."),
    ("rev_converted", 0, 16, cg((16, "M")), b"ACATCAATCTTACCAA", False, "..z...x.......z."),
    # at the contig's first base a reverse read pads the missing flank with N (tag.rs:167-173): G2's context is fine,
    # a read G at column 0 would look at "NN"; here column 0 is A -> '.'
    ("rev_at_contig_start", 0, 16, cg((3, "M")), b"ACG", False, "..Z"),
    # paired-end: read 2 on the forward strand is reverse-complemented (tag.rs:15-18, 141-144), read 1 forward is not
    ("paired_read2_fwd", 0, 128 | 1, cg((3, "M")), b"ACG", True, "..Z"),
    ("paired_read1_fwd", 0, 64 | 1, cg((3, "M")), b"ACG", True, ".Z."),
    ("paired_read1_rev", 0, 64 | 16 | 1, cg((3, "M")), b"ACG", True, "..Z"),
    ("paired_read2_rev", 0, 128 | 16 | 1, cg((3, "M")), b"ACG", True, ".Z."),
    # deletion right after a C (tag.rs:271-337): read A C [G deleted] T C -> the context skips the gap: C, then the
    # next two read-aligned reference bases T C -> "CTC" -> H.  Reference 0..4 = A C G T C, CIGAR 2M1D2M.
    # Columns: A . ; C H ; (gap: no letter) ; T . ; C4: next reference bases A G -> CAG -> X
    ("deletion_after_c", 0, 0, cg((2, "M"), (1, "D"), (2, "M")), b"ACTC", False, ".H.X"),
    # insertion right after a C: the reference column under the inserted base is '-', so the 3-letter context holds a
    # '-' -> unknown context U (tag.rs:375-383).  Reference A C | G T, read A C [T inserted] G T, CIGAR 2M1I2M.
    ("insertion_after_c", 0, 0, cg((2, "M"), (1, "I"), (2, "M")), b"ACTGT", False, ".U..."),
    # a soft clip is NOT walked (tag.rs:236 `_ => {}`): the 2 clipped bases stay at the front of the read string and
    # the 3M takes read[0..3] = "TTA" against reference 1..3 = C G T -> z . .   (a reference quirk, reproduced)
    ("soft_clip_shifts_the_read", 1, 0, cg((2, "S"), (3, "M")), b"TTACG", False, "z.."),
])
def test_hand_worked_records(name, pos, flag, cigar, seq, paired, want):
    got = pyoracle.tag_xm(pos, flag, cigar, seq, CONTIG, is_paired_end=paired)
    assert got is not None and got.decode() == want, (name, got)


def test_where_the_reference_panics():
    # a base the complement table does not hold ('=' from BAM code 0) on the reverse-complement path: HashMap index panics (tag.rs:24)
    assert pyoracle.tag_xm(0, 16, cg((3, "M")), b"A=G", CONTIG) is None
    # an alignment that ends beyond the contig by more than the two pad bases: padding[] index panics (tag.rs:170)
    assert pyoracle.tag_xm(len(CONTIG) - 1, 0, cg((5, "M")), b"AAAAA", CONTIG) is None
