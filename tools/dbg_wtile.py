"""debug aid: one FDRP pass of a reference fixture (or a small synthetic contig) through the one-pass tile form"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["METHEOR_FDRP_DEBUG"] = "1"
os.environ.setdefault("METHEOR_FDRP_WTILE", "1")
import numpy as np
import metheor_amd
from oracle import bamio, pyoracle
from tests import util, test_gpu_fdrp as T

eng = metheor_amd.Engine(0)
which = sys.argv[1] if len(sys.argv) > 1 else "fix1"
if which.startswith("fix"):
    reads, c = T.fixture(os.path.join(os.path.dirname(__file__), "..", "tests", "golden"), int(which[3:]))
    kw = dict(min_qual=0, min_depth=2, max_depth=40, min_overlap=4)
else:
    from metheor_amd import synth
    c = synth.make_contig(2, 300_000, 20_000, 0.0091, np.random.default_rng(77))
    reads = pyoracle.Reads.from_soa(*synth.to_oracle_soa(c))
    kw = dict(min_qual=10, min_depth=4, max_depth=40, min_overlap=35)
d = T.run_device(eng, [c], kw)
of = reads.fdrp(**kw)
print("device rows", len(d["pos"]), "oracle rows", len(of.pos))
print(d["pos"][:10], d["fdrp"][:10], d["qfdrp"][:10], d["n_reads"][:10])
print(of.pos[:10, 0], of.val[:10], of.cnt[:10, 0])
if len(d["pos"]) == len(of.pos):
    print(T.check(d, reads, kw))
