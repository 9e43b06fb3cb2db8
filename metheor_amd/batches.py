"""Contig dict (metheor_amd.synth / shard layout) -> metheor_amd.Batch, on the host or resident on a device.
Plumbing shared by bench.py, __graft_entry__.smoke(), the tools and the tests; nothing here computes a measure."""
import numpy as np

from . import shard
from .capi import Batch


def device_batch(c, region=None, device=None, rel16=False):
    """contig dict -> Batch (host numpy, or torch tensors on `device`)"""
    beg, end = region if region is not None else c.get("region", (0, c["length"]))
    rel = c["cpg_rel"].astype(np.uint16) if rel16 else c["cpg_rel"]
    arrs = dict(read_start=c["read_start"], read_end=c["read_end"], read_mapq=c["read_mapq"],
                cpg_off=c["cpg_off"], cpg_pos=c["cpg_pos"], cpg_rel=rel)
    if device is not None:
        import torch
        t = {}
        for k, a in arrs.items():
            a = np.ascontiguousarray(a)
            if a.dtype == np.uint32:
                t[k] = torch.from_numpy(a.view(np.int32)).to(device)
            elif a.dtype == np.uint16:
                t[k] = torch.from_numpy(a.view(np.int16)).to(device)
            else:
                t[k] = torch.from_numpy(a).to(device)
        arrs = t
    return Batch(c["tid"], beg, end, max_span=shard.max_span(c), **arrs)


def group_device_batches(engines, batches, lengths, max_span=150, vmax=(1 << 31) - (1 << 22)):
    """Device-resident per-contig Batches (torch tensors; tids ascending) -> the same reads as contig GROUPS (include/metheor_hip.h,
    "contig groups"): contigs packed greedily into virtual coordinate spaces of at most `vmax` positions, offsets a multiple of 4096
    with a gap of max_span + 1024 after every contig -- what mth_decoded_group does to a decoded stream, done here with torch for
    batches that never were one.  The groups are defined on every engine of `engines` (same order: same handles).  Returns the
    Batches (a group of one contig stays the contig's own Batch)."""
    import torch
    groups, cur, vlen = [], [], 0
    for b, ln in zip(batches, lengths):
        ext = ((int(ln) + max_span + 1024 + 4095) // 4096) * 4096
        if cur and (vlen + ext > vmax or sum(x[0].n_reads for x in cur) + b.n_reads >= (1 << 32) - 1 or sum(x[0].n_cpgs for x in cur) + b.n_cpgs >= (1 << 32)):
            groups.append(cur); cur, vlen = [], 0
        cur.append((b, int(ln), vlen)); vlen += ext
    groups.append(cur)
    out = []
    for g in groups:
        if len(g) == 1:
            out.append(g[0][0]); continue
        handles = {e.group_define([b.tid for b, _, _ in g], [vo for _, _, vo in g]) for e in engines}
        assert len(handles) == 1, "the engines must have the same groups defined so far"
        rs, re_, mq, off, pos, rel, base = [], [], [], [], [], [], 0
        for b, ln, vo in g:
            k = b.keep                                  # read_start, read_end, read_mapq, cpg_off, cpg_pos, cpg_rel (Batch.__init__'s order)
            rs.append(k[0] + vo); re_.append(k[1] + vo); mq.append(k[2])
            o = k[3].to(torch.int64) & 0xffffffff
            off.append(o[:-1] + base); base += int(o[-1].item())
            p = k[4].to(torch.int64) & 0xffffffff
            pos.append(((p & 0x7fffffff) + vo) | (p & 0x80000000))
            rel.append(k[5])
        off.append(torch.tensor([base], device=rs[0].device, dtype=torch.int64))
        as_i32 = lambda x: torch.where(x >= (1 << 31), x - (1 << 32), x).to(torch.int32).contiguous()       # u32 bit patterns in an int32 tensor
        out.append(Batch(handles.pop(), 0, g[-1][2] + g[-1][1], torch.cat(rs).to(torch.int32).contiguous(), torch.cat(re_).to(torch.int32).contiguous(),
                         torch.cat(mq).contiguous(), as_i32(torch.cat(off)), as_i32(torch.cat(pos)), torch.cat(rel).contiguous(), max_span=max_span))
    return out
