"""SAM text input / output and FASTA access of the host library (CPU): what the reference gets from htslib for `tag`
(tests/tag-cli.rs feeds it a .sam and compares SAM text) and for every measure given a .sam (bamutil.rs:4-11)."""
import os

import numpy as np
import pytest

from metheor_amd import hostapi
from oracle import bamio, pyoracle
from tests import tag_util


def test_sam_round_trip_is_the_identity_on_the_reference_fixture(golden_dir):
    # SAM text -> BAM records (mth_host_open on a .sam) -> SAM text: every line of the reference's 1000-read fixture unchanged
    path = os.path.join(golden_dir, "test.chr19.XM.sam")
    f = hostapi.BamFile(path)
    want = open(path, "rb").read().splitlines(keepends=True)
    hdr = [l for l in want if l.startswith(b"@")]
    assert f.header_text() == b"".join(hdr)
    assert f.refs == [("chr19", 58617616)]
    got = []
    for raw, off in f.windows():
        for k in range(len(off) - 1):
            got.append(f.sam_line(raw, int(off[k]), int(off[k + 1])))
    assert got == [l for l in want if not l.startswith(b"@")]


def test_sam_input_decodes_like_the_same_records_as_bam(golden_dir, tmp_path):
    # the measures' host decode over the SAM file == over a BAM holding the same records (independent Python writer)
    sam = os.path.join(golden_dir, "test.chr19.XM.sam")
    rec = bamio.read_sam(sam)
    bam = str(tmp_path / "same.bam")
    bamio.write_bam(bam, rec)
    a = hostapi.BamFile(sam).decode()
    b = hostapi.BamFile(bam).decode()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    o = pyoracle.Reads.decode(rec).soa()
    for k in ("tid", "start", "end", "mapq", "cpg_pos"):
        assert np.array_equal(a[k], o[k]), k


def test_sam_aux_types_and_odd_records(tmp_path):
    lines = ["@HD\tVN:1.6\tSO:coordinate", "@SQ\tSN:c1\tLN:1000", "@SQ\tSN:c2\tLN:500",
             "r1\t99\tc1\t10\t60\t2S5M1I3M2D4M\t=\t40\t50\tACGTNACGTNACGTN\tIIIIIIIIIIIIIII\tNM:i:-3\tXA:A:q\tXB:i:70000\tXC:i:-200\tXD:i:300\tXF:f:1.5\tXH:H:1AE3\tXZ:Z:a b\tBA:B:c,-1,2\tBB:B:S,1,65535\tBF:B:f,0.25,3",
             "r2\t147\tc1\t40\t0\t15M\tc2\t7\t-50\tacgtacgtacgtacg\t*",
             "r3\t4\t*\t0\t0\t*\t*\t0\t0\t*\t*"]
    p = tmp_path / "x.sam"
    p.write_text("\n".join(lines) + "\n")
    f = hostapi.BamFile(str(p))
    got = []
    for raw, off in f.windows():
        for k in range(len(off) - 1):
            got.append(f.sam_line(raw, int(off[k]), int(off[k + 1])).decode().rstrip("\n"))
    want = list(lines[3:])
    want[1] = want[1].replace("acgtacgtacgtacg", "ACGTACGTACGTACG")        # BAM stores 4-bit codes: case is not kept (htslib prints upper case)
    assert got == want
    assert f.sam_line(raw, int(off[0]), int(off[1]), xm=b"..Z").decode().endswith("\tBF:B:f,0.25,3\tXM:Z:..Z\n")


def test_not_sam_not_bam_is_still_an_open_error(tmp_path):
    p = tmp_path / "Cargo.toml"
    p.write_text("[package]\nname = \"metheor\"\n")
    with pytest.raises(hostapi.HostError) as e:      # tests/cli_error_handling.rs:214-227
        hostapi.BamFile(str(p))
    assert "Error opening BAM file" in str(e.value)
    q = tmp_path / "bad.sam"
    q.write_text("@HD\tVN:1.0\n@SQ\tSN:c1\tLN:10\nr1\t0\tc9\t1\t0\t1M\t*\t0\t0\tA\tI\n")
    with pytest.raises(hostapi.HostError) as e:
        hostapi.BamFile(str(q))
    assert "RNAME" in str(e.value)


def test_fasta_with_and_without_index(tmp_path):
    seq = bytes(np.random.default_rng(3).choice(list(b"ACGTNacgt"), size=1234).astype(np.uint8))
    for with_fai in (True, False):
        p = str(tmp_path / ("g%d.fa" % with_fai))
        tag_util.write_fasta(p, "chrA", seq, width=50, with_fai=False)
        with open(p, "ab") as fh:
            fh.write(b">chrB description text\nAC\nGT\n")
        if with_fai:
            open(p + ".fai", "w").write("chrA\t1234\t6\t50\t51\nchrB\t4\t%d\t2\t3\n" % (6 + 1234 + 25 + len(">chrB description text\n")))
        fa = hostapi.Fasta(p)
        assert fa.fetch("chrA", 1234) == seq                 # fetch_seq(name, 0, LN): inclusive end, clipped (tag.rs:424-428)
        assert fa.fetch("chrA", 9) == seq[:10]
        assert fa.fetch("chrB", 100) == b"ACGT"
        with pytest.raises(hostapi.HostError):
            fa.fetch("chrC", 10)
    with pytest.raises(hostapi.HostError) as e:              # tests/tag-cli.rs:42-58
        hostapi.Fasta(str(tmp_path / "no_such.fa"))
    assert "file not found" in str(e.value) and "no_such.fa" in str(e.value)


def test_reference_tinyref_fixture_layout(tmp_path):
    # the reference ships tests/tinyref.fa(.fai): one 51-base sequence "ref", index line "ref 51 5 51 52"
    p = str(tmp_path / "tinyref.fa")
    open(p, "w").write(">ref\n" + "CGGGGCGGGGCGCGCGGGGGCGCGCGCGGGGCGCGCGCGCGCGGGGGGGGG" + "\n")
    open(p + ".fai", "w").write("ref\t51\t5\t51\t52\n")
    assert hostapi.Fasta(p).fetch("ref", 51) == b"CGGGGCGGGGCGCGCGGGGGCGCGCGCGGGGCGCGCGCGCGCGGGGGGGGG"


def test_sam_line_longer_than_four_characters_per_record_byte(tmp_path):
    """a B:c / B:s array prints up to 5-7 characters per 1-2 byte element (",-128", ",-32768"): the formatted line
    outgrows the 4 x record bytes + 256 the callers used to size their buffer by.  mth_host_sam_format returns what it needs;
    hostapi.BamFile.sam_line (and the CLI's tag writer) ask again with that size."""
    arr_c = ",".join(["-128"] * 3000)
    arr_s = ",".join(["-32768"] * 1500)
    lines = ["@HD\tVN:1.6", "@SQ\tSN:c1\tLN:1000",
             "r1\t0\tc1\t10\t60\t4M\t*\t0\t0\tACGT\t*\tBA:B:c," + arr_c + "\tBS:B:s," + arr_s]
    p = tmp_path / "b.sam"
    p.write_text("\n".join(lines) + "\n")
    f = hostapi.BamFile(str(p))
    (raw, off), = f.windows()
    rec_bytes = int(off[1] - off[0])
    line = f.sam_line(raw, int(off[0]), int(off[1]), xm=b"....")
    assert len(line) > 4 * rec_bytes + 256 + 4          # the case the old bound missed
    assert line.decode() == lines[2] + "\tXM:Z:....\n"
