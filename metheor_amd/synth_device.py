"""metheor_amd.synth's generator with torch ops on the device: the same distributions -- CpG
sites at random, never adjacent, each with its own methylation level; error-free single-end reads, 50 % reverse strand, 5 % low
mapq, coordinate sorted -- straight into device-resident SoA, so that bench.py can put the whole of BASELINE config 3
(200 M reads over 24 hg38-sized contigs) into HBM in seconds instead of minutes of numpy.  Another random stream than
synth.make_contig (torch's generator, not numpy's): the parity tests keep the numpy generator, whose arrays the oracle reads.
Nothing here computes a measure."""
import math

import torch

from .capi import Batch
from .synth import HG38_LENGTHS


def make_sites(length, density, gen, device):
    """sorted CpG (C) positions in [1, length-2], gaps >= 2"""
    n_guess = int(length * density * 1.1) + 16
    u = torch.rand(n_guess, generator=gen, device=device, dtype=torch.float64)
    # geometric(p) on {1, 2, ...} by inversion, plus one: a CpG is two bases
    gaps = torch.floor(torch.log1p(-u) / math.log1p(-density)).to(torch.int64) + 2
    pos = torch.cumsum(gaps, 0)
    return pos[pos < length - 2]


def make_contig_tensors(tid, length, n_reads, density, gen, device, read_len=150, low_mapq_frac=0.05,
                        levels=((0.1, 0.3), (0.9, 0.7)), starts=None):
    """one contig's reads as device tensors (the SoA of include/metheor_hip.h); returns (tensor dict, info dict)"""
    sites = make_sites(length, density, gen, device)
    lv = torch.tensor([l for l, _ in levels], dtype=torch.float32, device=device)
    pw = torch.tensor([w for _, w in levels], dtype=torch.float64)
    cum = torch.cumsum(pw / pw.sum(), 0).to(device)
    site_level = lv[torch.searchsorted(cum, torch.rand(len(sites), generator=gen, device=device, dtype=torch.float64)).clamp_(max=len(levels) - 1)]
    if starts is None:
        starts = torch.randint(0, max(length - read_len, 1), (n_reads,), generator=gen, device=device, dtype=torch.int64)
        starts, _ = torch.sort(starts)
    n_reads = int(starts.shape[0])
    rev = (torch.rand(n_reads, generator=gen, device=device) < 0.5).to(torch.int64)
    low = torch.rand(n_reads, generator=gen, device=device) < low_mapq_frac
    mapq = torch.where(low, torch.randint(0, 10, (n_reads,), generator=gen, device=device), torch.full((n_reads,), 42, device=device)).to(torch.uint8)
    # forward reads call the C of a CpG (site p in [start, start + len)); reverse reads call the G (p + 1 in that range) and
    # report abspos - 1 = p (readutil.rs:338)
    lo = torch.searchsorted(sites, starts - rev, right=False)
    hi = torch.searchsorted(sites, starts + read_len - rev, right=False)
    cnt = hi - lo
    cpg_off = torch.zeros(n_reads + 1, dtype=torch.int64, device=device)
    torch.cumsum(cnt, 0, out=cpg_off[1:])
    total = int(cpg_off[-1].item())
    assert total < 2 ** 32
    read_of = torch.repeat_interleave(torch.arange(n_reads, device=device), cnt, output_size=total)
    site_idx = torch.arange(total, device=device) - (cpg_off[:-1] - lo)[read_of]
    pos = sites[site_idx]
    rel = (pos - starts[read_of] + rev[read_of]).to(torch.uint8 if read_len <= 256 else torch.int16)
    meth = torch.rand(total, generator=gen, device=device) < site_level[site_idx]
    cpg_pos = (pos | (meth.to(torch.int64) << 31)).to(torch.int32)       # wraps into the sign bit: the 32-bit pattern is what counts
    rs = starts.to(torch.int32)
    t = dict(tid=tid, length=int(length), max_span=read_len, read_start=rs, read_end=rs + (read_len - 1), read_mapq=mapq,
             cpg_off=cpg_off, cpg_pos=cpg_pos, cpg_rel=rel)
    info = dict(tid=tid, length=int(length), n_reads=n_reads, n_calls=total, n_sites_possible=int(len(sites)),
                n_mapq_ok_with_calls=int(((mapq >= 10) & (cnt > 0)).sum().item()), n_mapq_ok=int((mapq >= 10).sum().item()))
    return t, info


def to_batch(t, region=None):
    """tensor dict (make_contig_tensors / slice_region) -> Batch"""
    beg, end = region if region is not None else t.get("region", (0, t["length"]))
    return Batch(t["tid"], int(beg), int(end), read_start=t["read_start"], read_end=t["read_end"], read_mapq=t["read_mapq"],
                 cpg_off=t["cpg_off"].to(torch.int32), cpg_pos=t["cpg_pos"], cpg_rel=t["cpg_rel"], max_span=t["max_span"])


def make_contig(tid, length, n_reads, density, gen, device, **kw):
    """one contig's reads as a device-resident Batch; returns (Batch, info dict)"""
    t, info = make_contig_tensors(tid, length, n_reads, density, gen, device, **kw)
    return to_batch(t), info


def slice_region(t, beg, end, halo=None):
    """shard.slice_region on the device: the reads that can touch sites in [beg, end) -- start in [beg - halo, end] -- CSR rebased;
    returns (tensor dict, number of reads OWNED by the region: start in [beg, end))"""
    if halo is None:
        halo = t["max_span"]
    s = t["read_start"]
    key = torch.tensor([beg - halo, end + 1, beg, end], dtype=s.dtype, device=s.device)
    i0, i1, a0, a1 = [int(x) for x in torch.searchsorted(s, key, right=False).tolist()]
    o0, o1 = int(t["cpg_off"][i0].item()), int(t["cpg_off"][i1].item())
    out = dict(t)
    for k in ("read_start", "read_end", "read_mapq"):
        out[k] = t[k][i0:i1].contiguous()
    out["cpg_off"] = (t["cpg_off"][i0:i1 + 1] - o0).contiguous()
    out["cpg_pos"] = t["cpg_pos"][o0:o1].contiguous()
    out["cpg_rel"] = t["cpg_rel"][o0:o1].contiguous()
    out["region"] = (int(beg), int(end))
    return out, a1 - a0


def wgbs(n_reads=200_000_000, seed=2000, density=0.0091, contigs=None, device="cuda:0"):
    """BASELINE config 3: 24 hg38-sized contigs, reads spread by contig length; yields (Batch, info) per contig"""
    lens = HG38_LENGTHS if contigs is None else contigs
    tot = float(sum(lens))
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    for tid, ln in enumerate(lens):
        yield make_contig(tid, ln, int(round(n_reads * ln / tot)), density, gen, device)


def hotspots(n_windows=20000, window=1000, depth=50, density=0.08, seed=50, read_len=150, device="cuda:0"):
    """BASELINE config 4: 1-kbp windows at an exact depth"""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    stride = window + 2 * read_len + 404
    length = n_windows * stride + 1000
    per = int(depth * window / read_len)
    starts = (torch.arange(n_windows, device=device, dtype=torch.int64)[:, None] * stride + 300 +
              torch.randint(0, window - read_len, (n_windows, per), generator=gen, device=device, dtype=torch.int64)).reshape(-1)
    starts, _ = torch.sort(starts)
    return make_contig(0, length, len(starts), density, gen, device, read_len=read_len, starts=starts)
