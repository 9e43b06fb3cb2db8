"""debug: RRBS fixture through the stream kernel vs the oracle: which rows differ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import metheor_amd
from metheor_amd import PdrLpmdParams, shard
from oracle import bamio, pyoracle
from tests import util
g = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rec = bamio.read_sam(os.path.join(g, "test.chr19.XM.sam"))
reads = pyoracle.Reads.decode(rec)
c = util.contig_from_oracle_soa(reads.soa(), 0, rec.refs[0][1])
eng = metheor_amd.Engine(0)
kw = dict(min_depth=0, min_cpgs=0, min_qual=10)
eng.reset()
bt = util.device_batch(c)
print("n_reads", len(c["read_start"]), "max_span", shard.max_span(c))
eng.pdr_lpmd_accumulate(bt, PdrLpmdParams(**kw))
p = eng.pdr_fetch()
o = reads.pdr(**kw)
dp, op = set(p["pos"].tolist()), set(o.pos[:, 0].tolist())
print("missing on device", sorted(op - dp), "extra on device", sorted(dp - op))
st, off, pos = c["read_start"], c["cpg_off"], c["cpg_pos"]
for m in sorted(op - dp) + sorted(dp - op):
    for i in range(len(st)):
        ps = (pos[off[i]:off[i + 1]] & 0x7fffffff)
        if m in ps.tolist():
            print(" pos", m, "read", i, "start", st[i], "end", c["read_end"][i], "mapq", c["read_mapq"][i], "calls", ps.tolist(), "prev start", st[i - 1] if i else None, "next", st[i + 1] if i + 1 < len(st) else None)
