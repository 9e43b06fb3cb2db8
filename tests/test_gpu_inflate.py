"""GPU parity: BGZF / DEFLATE inflate on the device (mth_inflate.hip) against zlib, byte for byte, and the whole device
load path (inflate -> per-block record walk -> record + XM decode) against the oracle's decode."""
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import bamio, pyoracle
from tests.test_host_decode import _weird_records, same_soa

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import metheor_amd
    e = metheor_amd.Engine(0)
    yield e
    e.close()


def deflate_blocks(chunks, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    """raw DEFLATE payloads of the chunks, each followed by the gzip trailer (CRC32, ISIZE), concatenated
    -> (file bytes, coff, csize, isize)"""
    out, coff, csize, isize = bytearray(), [], [], []
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        z = co.compress(bytes(c)) + co.flush()
        coff.append(len(out)); csize.append(len(z)); isize.append(len(c))
        out += z + struct.pack("<II", zlib.crc32(bytes(c)) & 0xffffffff, len(c))
    return bytes(out), np.array(coff, np.uint64), np.array(csize, np.uint32), np.array(isize, np.uint32)


def payloads(rng):
    yield "random bytes", [rng.integers(0, 256, size=n, dtype=np.uint8).tobytes() for n in (1, 2, 100, 65280, 40000)]
    yield "text with long matches", [(b"the quick brown fox jumps over the lazy dog. " * 1500)[:65000], b"abcabcabc" * 7000]
    yield "runs (distance-1 overlapping copies)", [b"a" * 65280, b"ab" * 30000, bytes(1000), b"x" + b"y" * 258 + b"z" * 259 + b"w" * 600]
    dna = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=60000).tobytes()
    qual = (rng.integers(0, 40, size=60000) + 33).astype(np.uint8).tobytes()
    yield "sequence- and quality-like", [dna, qual, dna[:3000] + qual[:3000] + dna[:3000]]
    skew = rng.choice(np.arange(256, dtype=np.uint8), size=65000, p=np.r_[np.full(4, 0.2), np.full(252, 0.2 / 252)]).tobytes()
    yield "skewed symbols (long Huffman codes)", [skew, skew[:777]]
    geo = np.minimum(rng.geometric(0.35, size=65000) - 1, 255).astype(np.uint8).tobytes()     # code lengths beyond the 10-bit table
    yield "geometric symbols", [geo]


@pytest.mark.parametrize("level,strategy", [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                            (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE),
                                            (6, zlib.Z_FILTERED)])
def test_inflate_matches_zlib(eng, level, strategy):
    rng = np.random.default_rng(1000 + level * 10 + strategy)
    for name, chunks in payloads(rng):
        fb, coff, csize, isize = deflate_blocks(chunks, level, strategy)
        got = eng.bgzf_inflate(fb, coff, csize, isize).tobytes()
        want = b"".join(chunks)
        assert len(got) == len(want), name
        if got != want:
            bad = next(i for i in range(len(want)) if got[i] != want[i])
            raise AssertionError("%s: first difference at byte %d of %d" % (name, bad, len(want)))


def test_corrupt_payload_is_reported(eng):
    from metheor_amd import MthError
    data = b"some compressible text " * 2000
    fb, coff, csize, isize = deflate_blocks([data], 6)
    bad = bytearray(fb)
    bad[len(bad) // 2] ^= 0x55
    with pytest.raises(MthError) as e:
        got = eng.bgzf_inflate(bytes(bad), coff, csize, isize)
        assert got.tobytes() != data          # (a flipped bit that still inflates to the right length is possible in principle)
        raise MthError(-10, "wrong bytes")
    assert e.value.status == -10
    eng.reset()
    wrong = isize.copy(); wrong[0] -= 1       # ISIZE mismatch
    with pytest.raises(MthError):
        eng.bgzf_inflate(fb, coff, csize, wrong)
    eng.reset()
    assert eng.bgzf_inflate(fb, coff, csize, isize).tobytes() == data
    # a flipped DATA byte inside a stored block: the DEFLATE stream and ISIZE stay valid, only the CRC32 catches it
    rnd = np.random.default_rng(8).integers(0, 256, size=50_000, dtype=np.uint8).tobytes()
    fb0, c0, s0, i0 = deflate_blocks([rnd, b"tail" * 100], 0)
    bad0 = bytearray(fb0)
    bad0[1000] ^= 0x01
    with pytest.raises(MthError) as e:
        eng.bgzf_inflate(bytes(bad0), c0, s0, i0)
    assert e.value.status == -10 and "CRC32" in str(e.value)
    eng.reset()
    assert eng.bgzf_inflate(fb0, c0, s0, i0).tobytes() == rnd + b"tail" * 100


def block_table(path):
    """BGZF framing of a file -> (file bytes, coff, csize, isize of the blocks with data, header bytes of the BAM)"""
    fb = open(path, "rb").read()
    o, coff, csize, isize = 0, [], [], []
    while o < len(fb):
        xlen, = struct.unpack_from("<H", fb, o + 10)
        bsize = None
        e = 0
        while e + 4 <= xlen:
            si1, si2, slen = fb[o + 12 + e], fb[o + 13 + e], struct.unpack_from("<H", fb, o + 14 + e)[0]
            if si1 == 66 and si2 == 67:
                bsize, = struct.unpack_from("<H", fb, o + 16 + e)
            e += 4 + slen
        total = bsize + 1
        isz, = struct.unpack_from("<I", fb, o + total - 4)
        if isz:
            coff.append(o + 12 + xlen); csize.append(total - 12 - xlen - 8); isize.append(isz)
        o += total
    import gzip
    raw = gzip.decompress(fb)
    l_text, = struct.unpack_from("<i", raw, 4)
    h = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, h); h += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, h); h += 8 + l_name
    return fb, np.array(coff, np.uint64), np.array(csize, np.uint32), np.array(isize, np.uint32), h, raw


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_reference_fixtures_whole_path(eng, golden_dir, k):
    path = os.path.join(golden_dir, "test%d.bam" % k)       # written by samtools: records never straddle blocks
    fb, coff, csize, isize, hbytes, raw = block_table(path)
    assert eng.bgzf_inflate(fb, coff, csize, isize).tobytes() == raw
    eng.bgzf_decode(fb, coff, csize, isize, hbytes)
    same_soa(eng.decoded_fetch(), pyoracle.Reads.decode(bamio.read_bam(path)).soa())


def test_block_aligned_synthetic_bam_whole_path(eng, tmp_path):
    """our C++ writer lays blocks out like htslib (whole records per block): 20 000 reads, ~130 blocks"""
    from metheor_amd import hostapi, synth
    c = synth.make_contig(0, 120_000, 20_000, 0.03, np.random.default_rng(5))
    p = str(tmp_path / "syn.bam")
    hostapi.write_synthetic_bam(p, c, seed=3, threads=3)
    fb, coff, csize, isize, hbytes, raw = block_table(p)
    assert len(coff) > 100
    assert eng.bgzf_inflate(fb, coff, csize, isize).tobytes() == raw
    n, nc = eng.bgzf_decode(fb, coff, csize, isize, hbytes)
    assert n == 20_000
    same_soa(eng.decoded_fetch(), pyoracle.Reads.decode(bamio.read_bam(p)).soa())
    # appended in two halves of the block list (a file streamed in chunks)
    h = len(coff) // 2
    lo = int(coff[h]) - 18                                   # any byte range that covers the blocks will do
    eng.bgzf_decode(fb[:int(coff[h - 1] + csize[h - 1]) + 8], coff[:h], csize[:h], isize[:h], hbytes)
    eng.bgzf_decode(fb[lo:], coff[h:] - np.uint64(lo), csize[h:], isize[h:], 0, append=True)
    same_soa(eng.decoded_fetch(), pyoracle.Reads.decode(bamio.read_bam(p)).soa())


def test_straddling_records_are_refused(eng, tmp_path):
    """the pure-Python writer cuts blocks every 60 000 bytes: records straddle them -> MTH_ERR_UNALIGNED, nothing decoded"""
    from metheor_amd import MthError
    rec = _weird_records()
    p = str(tmp_path / "weird.bam")
    bamio.write_bam(p, rec)
    fb, coff, csize, isize, hbytes, raw = block_table(p)
    assert len(coff) >= 2
    assert eng.bgzf_inflate(fb, coff, csize, isize).tobytes() == raw      # the inflate itself is fine
    with pytest.raises(MthError) as e:
        eng.bgzf_decode(fb, coff, csize, isize, hbytes)
    assert e.value.status == -11
    eng.reset()


@pytest.mark.parametrize("seed", range(6))
def test_random_mixed_blocks(eng, seed):
    """many blocks in ONE call, each with its own random size, content model, zlib level and strategy (stored / fixed /
    dynamic mixed; short and long codes; literal-only and match-only blocks) -- byte-identical to the input"""
    rng = np.random.default_rng(7000 + seed)
    strategies = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]
    out, coff, csize, isize, want = bytearray(), [], [], [], []
    for _ in range(150):
        n = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 258, 259, 4096, 65279, 65280])) if rng.random() < 0.3 else int(rng.integers(1, 65281))
        kind = int(rng.integers(0, 7))
        if kind == 0:
            c = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        elif kind == 1:
            c = rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=n, p=[0.3, 0.2, 0.2, 0.29, 0.01]).tobytes()
        elif kind == 2:                                   # runs of random length and byte
            parts = []
            while sum(map(len, parts)) < n:
                parts.append(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 700)))
            c = b"".join(parts)[:n]
        elif kind == 3:                                   # repeats at assorted distances (1 .. 32767)
            base = rng.integers(0, 256, size=int(rng.integers(1, 40000)), dtype=np.uint8).tobytes()
            c = (base * (n // len(base) + 1))[:n]
        elif kind == 4:                                   # heavily skewed alphabet: code lengths up to 15
            p = 0.5 ** np.arange(1, 41); p /= p.sum()
            c = rng.choice(np.arange(40, dtype=np.uint8), size=n, p=p).tobytes()
        elif kind == 5:
            c = (b"XM:Z:" + b"." * 40 + b"z..Z" + b"\0" + bytes(rng.integers(33, 74, size=30, dtype=np.uint8))) * (n // 80 + 1)
            c = c[:n]
        else:
            c = bytes(n)
        co = zlib.compressobj(int(rng.choice([0, 1, 4, 6, 9])), zlib.DEFLATED, -15, int(rng.integers(1, 10)), int(rng.choice(strategies)))
        z = co.compress(c) + co.flush()
        coff.append(len(out)); csize.append(len(z)); isize.append(len(c)); want.append(c)
        out += z + struct.pack("<II", zlib.crc32(c) & 0xffffffff, len(c)) + bytes(int(rng.integers(0, 5)))   # arbitrary payload alignment
    got = eng.bgzf_inflate(bytes(out), np.array(coff, np.uint64), np.array(csize, np.uint32), np.array(isize, np.uint32)).tobytes()
    assert got == b"".join(want)
