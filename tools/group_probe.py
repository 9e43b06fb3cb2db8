"""Feasibility of contig groups: config 3's 24 per-contig batches against the same reads laid out in a few virtual contigs
(positions + offset, gap 4096): pass times and row counts.  python tools/group_probe.py [reads] [max virtual length]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, metheor_amd
from metheor_amd import synth_device, capi
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000_000
vmax = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 31) - (1 << 22)
dev = torch.device("cuda:0")
eng = metheor_amd.Engine(0)
res = [(b, inf) for b, inf in synth_device.wgbs(n_reads=n_reads, device=dev)]
torch.cuda.synchronize()
# groups
groups, cur, vlen = [], [], 0
for b, inf in res:
    ext = ((inf["length"] + 4096 + 4095) // 4096) * 4096
    if cur and vlen + ext > vmax:
        groups.append(cur); cur, vlen = [], 0
    cur.append((b, inf, vlen)); vlen += ext
groups.append(cur)
gb = []
for g in groups:
    rs, re_, mq, off, pos, rel = [], [], [], [], [], []
    base = 0
    for b, inf, vo in g:
        k = b.keep          # read_start, read_end, read_mapq, cpg_off, cpg_pos, cpg_rel (torch tensors, in that order)
        rs.append(k[0] + vo); re_.append(k[1] + vo); mq.append(k[2])
        o = k[3].to(torch.int64)
        off.append(o[:-1] + base); base += int(o[-1].item())
        p = k[4].to(torch.int64) & 0xffffffff
        pos.append((((p & 0x7fffffff) + vo) | (p & 0x80000000)).to(torch.int64))
        rel.append(k[5])
    off.append(torch.tensor([base], device=dev, dtype=torch.int64))
    cat = lambda xs: torch.cat(xs).contiguous()
    off32 = cat(off)
    assert base < (1 << 32)
    off32 = torch.where(off32 >= (1 << 31), off32 - (1 << 32), off32).to(torch.int32)
    p64 = cat(pos)
    p32 = torch.where(p64 >= (1 << 31), p64 - (1 << 32), p64).to(torch.int32)
    last = g[-1]
    gb.append(capi.Batch(-2 - len(gb), 0, last[2] + last[1]["length"], cat(rs).to(torch.int32), cat(re_).to(torch.int32), cat(mq), off32, p32, cat(rel), max_span=150))
torch.cuda.synchronize()
print("groups", [(len(g), b.n_reads) for g, b in zip(groups, gb)], flush=True)
P0 = metheor_amd.PdrLpmdParams()
passes = {"pdr+lpmd": lambda b: eng.pdr_lpmd_accumulate(b, P0), "me/pm": lambda b: eng.quartet_accumulate(b),
          "mhl": lambda b: eng.mhl_accumulate(b), "fdrp+qfdrp": lambda b: eng.fdrp_accumulate(b), "pairs": lambda b: eng.lpmd_pairs_accumulate(b)}
counts = {"pdr+lpmd": lambda: eng.pdr_count(), "me/pm": lambda: len(eng.quartet_fetch(min_depth=10)["me"]), "mhl": lambda: len(eng.mhl_fetch()["pos"]),
          "fdrp+qfdrp": lambda: len(eng.fdrp_fetch()["pos"]), "pairs": lambda: len(eng.lpmd_pairs_fetch()["pos1"])}
for name, fn in passes.items():
    line = [name]
    for label, bs in (("per-contig", [b for b, _ in res]), ("grouped", gb)):
        best = None
        for _ in range(3):
            eng.reset(); eng.sync()
            t0 = time.perf_counter()
            for b in bs:
                fn(b)
            eng.sync()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        line += [label, round(best * 1e3, 3), "rows", counts[name]()]
    print(*line, flush=True)
