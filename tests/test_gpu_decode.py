"""GPU parity: BAM record + XM decode on the device (mth_decode.hip) against the oracle's decode
(oracle/bamio.py pure-Python loader + orc_decode) -- bit-exact SoA -- and, end to end, measures computed from
the device-decoded arrays (no host SoA in between) against the oracle's measures on the same records."""
import gzip
import os
import struct

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests import test_gpu_pdr_lpmd as T_pdr
from tests.test_host_decode import KEYS, _weird_records, same_soa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


def record_stream(path):
    """BGZF-inflate a BAM file (concatenated gzip members), skip the header, walk the block_size fields:
    -> (refs, record bytes as uint8 array, uint64 offsets of the records (n + 1))"""
    raw = gzip.decompress(open(path, "rb").read())
    assert raw[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", raw, 4)
    o = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, o)
    o += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, o)
        name = raw[o + 4:o + 4 + l_name - 1].decode()
        l_ref, = struct.unpack_from("<i", raw, o + 4 + l_name)
        refs.append((name, l_ref))
        o += 8 + l_name
    body = np.frombuffer(raw, np.uint8)[o:]
    offs = [0]
    while offs[-1] < len(body):
        bs, = struct.unpack_from("<i", raw, o + offs[-1])
        offs.append(offs[-1] + 4 + bs)
    assert offs[-1] == len(body)
    return refs, np.ascontiguousarray(body), np.array(offs, np.uint64)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_reference_fixtures(eng, golden_dir, k):
    path = os.path.join(golden_dir, "test%d.bam" % k)
    refs, body, offs = record_stream(path)
    n, nc = eng.decode_records(body, offs)
    want = pyoracle.Reads.decode(bamio.read_bam(path)).soa()
    assert n == len(want["tid"]) and nc == len(want["cpg_pos"])
    same_soa(eng.decoded_fetch(), want)


def test_real_rrbs_reads(eng, golden_dir, tmp_path):
    rec = bamio.read_sam(os.path.join(golden_dir, "test.chr19.XM.sam"))
    p = str(tmp_path / "rrbs.bam")
    bamio.write_bam(p, rec)
    refs, body, offs = record_stream(p)
    assert eng.decode_records(body, offs)[0] == 1000
    got = eng.decoded_fetch()
    same_soa(got, pyoracle.Reads.decode(rec).soa())
    assert (got["fwd"] == 0).sum() == 676


@pytest.mark.parametrize("device_mem", [False, True])
def test_cigar_and_strand_rules(eng, tmp_path, device_mem):
    """indels, clips, ref-skips, =/X, paired flags, XM shorter than the query, an unaligned record, three contigs,
    records straddling BGZF blocks (the generator of tests/test_host_decode.py)"""
    import torch
    rec = _weird_records()
    p = str(tmp_path / "weird.bam")
    bamio.write_bam(p, rec)
    refs, body, offs = record_stream(p)
    if device_mem:
        body, offs = torch.from_numpy(body.copy()).cuda(), torch.from_numpy(offs.view(np.int64).copy()).cuda()
    eng.decode_records(body, offs)
    got = eng.decoded_fetch()
    same_soa(got, pyoracle.Reads.decode(rec).soa())
    assert (got["start"] == -1).any() and (got["fwd"] == 0).any() and (got["fwd"] == 1).any()


def test_errors_and_empty(eng, golden_dir):
    from metheor_amd import MthError
    assert eng.decode_records(np.zeros(0, np.uint8), np.zeros(1, np.uint64)) == (0, 0)
    assert len(eng.decoded_fetch()["tid"]) == 0
    refs, body, offs = record_stream(os.path.join(golden_dir, "test1.bam"))
    # a record without XM:Z: rename the tag of the first record (readutil.rs:46 panics there)
    b2 = body.copy()
    at = bytes(b2[: int(offs[1])]).find(b"XMZ")
    assert at > 0
    b2[at + 1] = ord("Q")
    with pytest.raises(MthError) as e:
        eng.decode_records(b2, offs)
    assert e.value.status == -10 and "XM" in str(e.value)
    eng.reset()
    # offsets that do not match the block_size fields
    bad = offs.copy()
    bad[1] += 1
    with pytest.raises(MthError) as e:
        eng.decode_records(body, bad)
    assert e.value.status == -10
    eng.reset()
    assert eng.decode_records(body, offs)[0] == len(offs) - 1      # the context is usable again


def test_decode_to_measures_on_device(eng, tmp_path):
    """raw record stream -> device decode -> per-contig device batches -> PDR + LPMD, never through a host SoA"""
    from metheor_amd import PdrLpmdParams, synth
    from tests import util
    rng = np.random.default_rng(91)
    cs = [synth.make_contig(0, 90_000, 14_000, 0.03, rng), synth.make_contig(1, 60_000, 9_000, 0.03, rng, read_len=100)]
    r0, r1 = util.contig_to_records(cs[0], "c0"), util.contig_to_records(cs[1], "c1")
    recs = bamio.Records([("c0", 90_000), ("c1", 60_000)], np.concatenate([r0.tid, r1.tid + 1]), np.concatenate([r0.pos, r1.pos]),
                         np.concatenate([r0.flag, r1.flag]), np.concatenate([r0.mapq, r1.mapq]), r0.cigars + r1.cigars, r0.xms + r1.xms)
    p = str(tmp_path / "syn.bam")
    bamio.write_bam(p, recs)
    reads = pyoracle.Reads.decode(recs)
    refs, body, offs = record_stream(p)
    n, nc = eng.decode_records(body, offs)
    dec = eng.decoded_fetch()
    same_soa(dec, reads.soa())
    eng.reset()
    kw = dict(min_depth=5, min_cpgs=2, min_qual=10)
    tids = dec["tid"]
    for t, (_, length) in enumerate(refs):
        r0, r1 = int(np.searchsorted(tids, t, "left")), int(np.searchsorted(tids, t, "right"))
        eng.pdr_lpmd_accumulate(eng.decoded_batch(r0, r1, t, 0, length), PdrLpmdParams(**kw))
    T_pdr.check_against_oracle(eng.pdr_fetch(), eng.lpmd_global(), reads, kw, dict())


def test_cpg_set_filter_on_device(eng, tmp_path):
    """--cpg-set (filter_isin, readutil.rs:87-95) applied by the decode kernel: same SoA as the oracle's filtered decode,
    relpos preserved; an empty set drops every call; removing the filter restores the full decode"""
    rec = _weird_records()
    p = str(tmp_path / "weird.bam")
    bamio.write_bam(p, rec)
    refs, body, offs = record_stream(p)
    full = pyoracle.Reads.decode(rec).soa()
    pos = full["cpg_pos"] & 0x7fffffff
    tid_of_call = np.repeat(full["tid"], np.diff(full["cpg_off"]).astype(np.int64))
    pick = np.random.default_rng(3).random(len(pos)) < 0.3
    sites = sorted(set(zip(tid_of_call[pick].tolist(), pos[pick].tolist())))
    try:
        eng.decode_set_cpg_filter(sites)
        eng.decode_records(body, offs)
        got = eng.decoded_fetch()
        same_soa(got, pyoracle.Reads.decode(rec, cpg_set=sites).soa())
        assert 0 < len(got["cpg_pos"]) < len(full["cpg_pos"]) and got["cpg_rel"].max() > 10
        eng.decode_set_cpg_filter([])
        eng.decode_records(body, offs)
        assert len(eng.decoded_fetch()["cpg_pos"]) == 0
    finally:
        eng.decode_set_cpg_filter(None)
    eng.decode_records(body, offs)
    same_soa(eng.decoded_fetch(), full)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_aux_fields(eng, tmp_path, seed):
    """aux fields of every BAM type (A c C s S i I f Z H, B arrays) before and after XM:Z: the device aux scanner steps over them"""
    from tests.test_host_decode import random_aux_fields
    rec = _weird_records()
    rec.aux_extra = random_aux_fields(np.random.default_rng(500 + seed), len(rec))
    p = str(tmp_path / "aux.bam")
    bamio.write_bam(p, rec)
    refs, body, offs = record_stream(p)
    eng.decode_records(body, offs)
    same_soa(eng.decoded_fetch(), pyoracle.Reads.decode(rec).soa())


@pytest.mark.timeout(120)
def test_crafted_aux_is_rejected_not_looped(eng, golden_dir, tmp_path):
    """a B-array count that wraps a 32-bit cursor back onto its own tag must end the record's scan (MTH_ERR_FORMAT), not spin
    the decode kernel; same for counts running past the record and an unknown sub-type (htslib rejects them: the reference
    panics with the XM text, readutil.rs:46-48)"""
    from metheor_amd import MthError
    from tests.test_host_decode import crafted_aux_cases
    for case, blob in sorted(crafted_aux_cases().items()):
        rec = bamio.read_bam(os.path.join(golden_dir, "test1.bam"))
        rec.aux_extra = [(b"NMC\x00", b"XRZCT\0")] * len(rec)
        rec.aux_extra[2] = (blob, b"")
        p = str(tmp_path / ("crafted_%s.bam" % case))
        bamio.write_bam(p, rec)
        refs, body, offs = record_stream(p)
        with pytest.raises(MthError) as e:
            eng.decode_records(body, offs)
        assert e.value.status == -10, case
        eng.reset()
    refs, body, offs = record_stream(os.path.join(golden_dir, "test1.bam"))
    assert eng.decode_records(body, offs)[0] == len(offs) - 1      # the context is usable again


def test_hand_worked_cigar_and_strand_rules(eng, tmp_path):
    """the device decoder against expectations worked out by hand from the SAM spec and readutil.rs (tests/test_host_decode.py)"""
    from tests.test_host_decode import check_hand_worked, hand_worked_records
    rec, expected = hand_worked_records()
    p = str(tmp_path / "hand.bam")
    bamio.write_bam(p, rec)
    refs, body, offs = record_stream(p)
    eng.decode_records(body, offs)
    check_hand_worked(eng.decoded_fetch(), expected)


def test_device_sort_of_a_shuffled_stream(eng, tmp_path):
    """mth_decoded_sort: the decoded SoA of a shuffled BAM, re-ordered by (tid, start) on the device, equals the stable numpy sort of the
    same arrays (reads AND their calls), the run finder then sees one sorted run per contig, and LPMD over the device batches equals the
    oracle streaming the shuffled file (lpmd.rs:175-200 does not care about the order)"""
    from metheor_amd import PdrLpmdParams, synth
    from tests import util
    rng = np.random.default_rng(92)
    cs = [synth.make_contig(0, 70_000, 11_000, 0.03, rng), synth.make_contig(1, 40_000, 6_000, 0.03, rng, read_len=100)]
    r0, r1 = util.contig_to_records(cs[0], "c0"), util.contig_to_records(cs[1], "c1")
    n = 17_000
    perm = rng.permutation(n)
    tid = np.concatenate([r0.tid, r1.tid + 1])[perm]
    recs = bamio.Records([("c0", 70_000), ("c1", 40_000)], tid, np.concatenate([r0.pos, r1.pos])[perm], np.concatenate([r0.flag, r1.flag])[perm],
                         np.concatenate([r0.mapq, r1.mapq])[perm], [(r0.cigars + r1.cigars)[i] for i in perm], [(r0.xms + r1.xms)[i] for i in perm])
    p = str(tmp_path / "shuf.bam")
    bamio.write_bam(p, recs)
    reads = pyoracle.Reads.decode(recs)
    refs, body, offs = record_stream(p)
    eng.decode_records(body, offs)
    before = eng.decoded_fetch()
    t0, _, _, fl = eng.decoded_contigs()
    assert len(t0) > 100 and (fl & 4 or len(t0) > 2)
    eng.decoded_sort()
    got = eng.decoded_fetch()
    order = np.lexsort((before["start"], before["tid"]))           # stable: ties keep the file order, as the radix sort does
    for k in ("tid", "start", "end", "mapq", "fwd"):
        assert (got[k] == before[k][order]).all(), k
    cnt = np.diff(before["cpg_off"].astype(np.int64))
    assert (np.diff(got["cpg_off"].astype(np.int64)) == cnt[order]).all()
    src = np.concatenate([np.arange(before["cpg_off"][i], before["cpg_off"][i + 1]) for i in order]).astype(np.int64)
    assert (got["cpg_pos"] == before["cpg_pos"][src]).all() and (got["cpg_rel"] == before["cpg_rel"][src]).all()
    tids, rb, re_, fl = eng.decoded_contigs()
    assert tids.tolist() == [0, 1] and fl == 0 and int(re_[1]) == n
    eng.reset()
    for t, (_, length) in enumerate(refs):
        eng.pdr_lpmd_accumulate(eng.decoded_batch(int(rb[t]), int(re_[t]), t, 0, length), PdrLpmdParams(want_pdr=False))
    l, o = eng.lpmd_global(), reads.lpmd()
    assert all(l[k] == o[k] for k in ("n_concordant", "n_discordant", "n_read", "n_valid_read"))
