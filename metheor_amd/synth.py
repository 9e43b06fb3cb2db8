"""Seeded synthetic Bismark-like WGBS input, generated straight into the SoA of include/metheor_hip.h.

Generators follow SURVEY.md section 8(d): error-free 150-bp single-end reads over a contig whose CpG
sites are placed at random (never adjacent), each site with its own methylation level.  Used by
bench.py and the tests (numpy only; nothing here computes a measure).
"""
import numpy as np

HG38_LENGTHS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
                145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
                101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468,
                156040895, 57227415]
HG38_NAMES = ["chr%d" % i for i in range(1, 23)] + ["chrX", "chrY"]
CHR19_LEN = 58617616


def make_sites(length, density, rng):
    """sorted CpG (C) positions in [1, length-2], gaps >= 2 (a CpG is two bases)"""
    n_guess = int(length * density * 1.1) + 16
    gaps = rng.geometric(density, size=n_guess).astype(np.int64) + 1
    pos = np.cumsum(gaps)
    pos = pos[pos < length - 2]
    return pos.astype(np.int32)


def make_contig(tid, length, n_reads, density, rng, read_len=150, low_mapq_frac=0.05,
                levels=((0.1, 0.3), (0.9, 0.7)), sites=None, starts=None):
    """one contig's reads as SoA dict (host numpy arrays), coordinate sorted"""
    if sites is None:
        sites = make_sites(length, density, rng)
    lv = np.array([l for l, _ in levels], dtype=np.float32)
    pw = np.array([w for _, w in levels], dtype=np.float64)
    site_level = lv[rng.choice(len(lv), size=len(sites), p=pw / pw.sum())]
    if starts is None:
        starts = np.sort(rng.integers(0, max(length - read_len, 1), size=n_reads, dtype=np.int64)).astype(np.int32)
    n_reads = len(starts)
    rev = (rng.random(n_reads) < 0.5)
    mapq = np.where(rng.random(n_reads) < low_mapq_frac, rng.integers(0, 10, size=n_reads), 42).astype(np.uint8)
    # forward reads call the C of a CpG (site position p in [start, start+len) );
    # reverse reads call the G (p+1 in [start, start+len)) and report abspos-1 = p (readutil.rs:338)
    s64 = starts.astype(np.int64)
    lo = np.searchsorted(sites, s64 - rev, side="left")
    hi = np.searchsorted(sites, s64 + read_len - rev, side="left")
    cnt = (hi - lo).astype(np.int64)
    cpg_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(cnt, out=cpg_off[1:])
    total = int(cpg_off[-1])
    read_of = np.repeat(np.arange(n_reads, dtype=np.int64), cnt)
    site_idx = np.arange(total, dtype=np.int64) - np.repeat(cpg_off[:-1] - lo, cnt)
    pos = sites[site_idx].astype(np.int64)
    rel = pos - s64[read_of] + rev[read_of]
    meth = rng.random(total, dtype=np.float32) < site_level[site_idx]
    cpg_pos = (pos.astype(np.uint32) | (meth.astype(np.uint32) << np.uint32(31))).astype(np.uint32)
    assert total < 2 ** 32
    return dict(tid=tid, length=int(length), read_start=starts, read_end=(starts + (read_len - 1)).astype(np.int32),
                read_mapq=mapq, read_fwd=(~rev).astype(np.uint8), cpg_off=cpg_off.astype(np.uint32),
                cpg_pos=cpg_pos, cpg_rel=rel.astype(np.uint8 if read_len <= 256 else np.uint16),
                n_sites_possible=len(sites))


def chr19_10m(n_reads=10_000_000, seed=1234, site_seed=19, density=0.02):
    """BASELINE config 2: S-chr19-10M (~25.6x, ~3.0 CpG calls/read at the default size)"""
    sites = make_sites(CHR19_LEN, density, np.random.default_rng(site_seed))
    return make_contig(0, CHR19_LEN, n_reads, density, np.random.default_rng(seed), sites=sites)


def wgbs(n_reads=200_000_000, seed=2000, density=0.0091, contigs=None):
    """BASELINE config 3/5: 24 hg38-sized contigs, reads spread by contig length; yields per-contig SoA"""
    lens = HG38_LENGTHS if contigs is None else contigs
    tot = float(sum(lens))
    rng = np.random.default_rng(seed)
    for tid, ln in enumerate(lens):
        yield make_contig(tid, ln, int(round(n_reads * ln / tot)), density, rng)


def wgbs_small(n_reads=2_000_000, seed=2000, density=0.0091, dense_frac=0.85, dense_span_frac=0.002, contigs=None):
    """config 3 / 5 at a size a parity test can check against the oracle: the same 24 hg38-sized contigs (same lengths,
    so the same coordinates, tile counts and contig changes as wgbs()), reads spread by contig length; inside a contig
    dense_frac of the reads fall into a few windows that together cover dense_span_frac of it (WGBS-like depth there)
    and the rest uniformly (almost every tile empty or with one read: the sparse-tile paths)."""
    lens = HG38_LENGTHS if contigs is None else contigs
    tot = float(sum(lens))
    rng = np.random.default_rng(seed)
    out = []
    for tid, ln in enumerate(lens):
        n = int(round(n_reads * ln / tot))
        nd = int(n * dense_frac)
        hi = max(ln - 150, 1)
        nwin = 4
        wlen = max(int(ln * dense_span_frac / nwin), 1000)
        w0 = np.sort(rng.integers(0, max(hi - wlen, 1), size=nwin))
        w0[-1] = max(hi - wlen, 0)                       # one window ends at the contig's last base
        dense = (w0[rng.integers(0, nwin, size=nd)] + rng.integers(0, wlen, size=nd)).astype(np.int64)
        starts = np.sort(np.concatenate([dense, rng.integers(0, hi, size=n - nd)])).astype(np.int32)
        out.append(make_contig(tid, ln, n, density, rng, starts=starts))
    return out


def hotspots(n_windows=20000, window=1000, depth=50, density=0.08, seed=50, read_len=150):
    """BASELINE config 4: 1-kbp windows at an exact depth (stresses the O(d^2) read-pair tile)"""
    rng = np.random.default_rng(seed)
    stride = window + 2 * read_len + 404
    length = n_windows * stride + 1000
    per = int(depth * window / read_len)
    starts = (np.arange(n_windows, dtype=np.int64)[:, None] * stride + 300 +
              rng.integers(0, window - read_len, size=(n_windows, per))).reshape(-1)
    starts = np.sort(starts).astype(np.int32)
    return make_contig(0, length, len(starts), density, rng, read_len=read_len, starts=starts)


def to_oracle_soa(c):
    """the dict -> argument tuple of oracle.pyoracle.Reads.from_soa"""
    n = len(c["read_start"])
    return (np.full(n, c["tid"], np.int32), c["read_start"], c["read_end"], c["read_mapq"], c["read_fwd"],
            c["cpg_off"].astype(np.uint64), c["cpg_pos"], c["cpg_rel"].astype(np.uint16))


def concat_oracle_soa(contigs):
    parts = [to_oracle_soa(c) for c in contigs]
    off = [np.uint64(0)]
    cpg_off = []
    for p in parts:
        cpg_off.append(p[5][:-1] + off[-1])
        off.append(off[-1] + p[5][-1])
    cpg_off.append(np.array([off[-1]], dtype=np.uint64))
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]), np.concatenate([p[3] for p in parts]),
            np.concatenate([p[4] for p in parts]), np.concatenate(cpg_off),
            np.concatenate([p[6] for p in parts]), np.concatenate([p[7] for p in parts]))
