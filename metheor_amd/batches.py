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
