"""The BAM index (.bai) path, SURVEY 8(f).2 (CPU part): the reference's fixtures ship tests/test{1..6}.bam.bai (samtools-made,
copied to tests/golden/ as data) although its reader never opens them.
* oracle/bamio.py's independent Python builder must reproduce every fixture (bins, chunks, linear index) from the BAM alone
  -> the builder is pinned against real samtools output and can then index the synthetic BAMs of the other tests;
* the product's planner (libmetheor_host: mth_host_plan_region = C++ .bai parser + reg2bins + linear-index query) must hand
  out blocks that contain EVERY record overlapping the region and its halo, for the fixtures and for random regions of a
  multi-contig BAM, and agree with the Python query on the virtual-offset range."""
import os

import numpy as np
import pytest

from metheor_amd import hostapi, synth
from oracle import bamio


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_python_builder_reproduces_the_reference_fixtures(golden_dir, k):
    bam = os.path.join(golden_dir, "test%d.bam" % k)
    want = bamio.read_bai(bam + ".bai")
    assert bamio.build_bai(bam) == want
    assert len(want) == 1                                          # chr1 only
    if k != 5:
        assert want[0]["bins"] and want[0]["ioffset"]


@pytest.mark.parametrize("k", [1, 4, 6])
def test_planner_on_the_reference_fixtures(golden_dir, k):
    bam = os.path.join(golden_dir, "test%d.bam" % k)
    f = hostapi.BamFile(bam)
    bz = f.bgzf_blocks()
    p = f.plan_region(0, 0, 100, halo_bp=0)                         # picks up tests/golden/testK.bam.bai by itself
    assert (p["tid_beg"], p["pos_beg"], p["tid_end"], p["pos_end"]) == (0, 0, 0, 100)
    assert p["block_beg"] == 0 and p["first_byte"] == bz["header_bytes"] and p["block_end"] >= 1   # the header's block comes along
    assert f.plan_region(0, 5_000_000, 5_000_100, halo_bp=0)["block_beg"] == f.plan_region(0, 5_000_000, 5_000_100, halo_bp=0)["block_end"]   # nothing there
    with pytest.raises(hostapi.HostError) as e:
        f.plan_region(0, 0, 10, bai=os.path.join(golden_dir, "no_such.bai"))
    assert "BAM index" in str(e.value)


def test_planner_covers_every_overlapping_record(tmp_path):
    rng = np.random.default_rng(41)
    names = ["rA", "rEmpty", "rB", "rC"]
    cs = [synth.make_contig(0, 900_000, 30_000, 0.02, rng), synth.make_contig(2, 2_500_000, 60_000, 0.02, rng),
          synth.make_contig(3, 300_000, 8_000, 0.02, rng)]
    bam = str(tmp_path / "multi.bam")
    hostapi.write_synthetic_bam_multi(bam, cs, names, seed=3, threads=4)
    refs = bamio.write_bai(bam)
    recs, _ = bamio.bam_record_offsets(bam)
    f = hostapi.BamFile(bam)
    bz = f.bgzf_blocks()
    assert len(bz["coff"]) > 200
    tids = np.array([r[0] for r in recs]); beg = np.array([r[1] for r in recs]); end = np.array([r[2] for r in recs])
    blk = np.array([r[5] for r in recs])                             # BGZF block of each record (the table omits no data block here)
    lens = {0: 900_000, 2: 2_500_000, 3: 300_000}
    for _ in range(300):
        t = int(rng.choice([0, 2, 3]))
        b = int(rng.integers(0, lens[t]))
        e = int(min(lens[t], b + rng.choice([1, 50, 5_000, 200_000, 3_000_000])))
        halo = int(rng.choice([0, 300, 65_536]))
        p = f.plan_region(t, b, e, halo_bp=halo)
        need = np.nonzero((tids == t) & (end > max(0, b - halo)) & (beg < e + 1))[0]          # records overlapping [b - halo, e]
        if len(need) == 0:
            continue
        assert p["block_end"] > p["block_beg"]
        assert p["block_beg"] <= blk[need].min() and blk[need].max() < p["block_end"], (t, b, e, halo, p)
        q = bamio.bai_query(refs, t, max(0, b - halo), e + 1)
        assert q is not None and (q[0] >> 16) >= (0 if p["block_beg"] == 0 else bz["coff"][p["block_beg"]] - 18)
    # an empty contig, and a region right of every read
    p = f.plan_region(1, 0, 1000)
    assert p["block_beg"] == p["block_end"]
    # the index of another file is refused
    other = str(tmp_path / "other.bam")
    hostapi.write_synthetic_bam(other, cs[0], contig="rA", seed=1)
    bamio.write_bai(other)
    with pytest.raises(hostapi.HostError) as ex:
        f.plan_region(0, 0, 10, bai=other + ".bai")
    assert "does not belong" in str(ex.value)
