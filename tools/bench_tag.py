"""`metheor tag` end to end on a generated BAM (no XM tags) + FASTA: whole-process wall time, reads/s.
Usage (GPU box): python tools/bench_tag.py [reads] [reps]"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metheor_amd import hostapi
from tests import tag_util
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(11)
L = 5_000_000
contig = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L)
fa = "/dev/shm/bench_tag.fa"
tag_util.write_fasta(fa, "chrT", contig.tobytes())
starts = np.sort(rng.integers(0, L - 150, size=n))
sam = "/dev/shm/bench_tag.sam"
with open(sam, "w") as fh:
    fh.write("@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrT\tLN:%d\n" % L)
    q = "I" * 150
    for i, s in enumerate(starts):
        seq = contig[s:s + 150].copy()
        conv = rng.random(150) < 0.7
        if i & 1:
            seq[(seq == ord("G")) & conv] = ord("A"); flag = 16
        else:
            seq[(seq == ord("C")) & conv] = ord("T"); flag = 0
        fh.write("r%d\t%d\tchrT\t%d\t40\t150M\t*\t0\t0\t%s\t%s\tNM:i:0\n" % (i, flag, s + 1, seq.tobytes().decode(), q))
f = hostapi.BamFile(sam)
bam = "/dev/shm/bench_tag.bam"
open(bam, "wb").write(open(f.staged_path(), "rb").read())
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metheor_amd", "metheor")
for src in (bam, sam):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "tag", "-i", src, "-o", "/dev/shm/bench_tag.out.sam", "-g", fa], capture_output=True, text=True, env=dict(os.environ, METHEOR_TIMING="1"))
        ts.append(time.perf_counter() - t0)
        assert r.returncode == 0, r.stderr
    print("%s input: %d reads, best %.3f s = %.2f M reads/s, output %d MB" % (src.rsplit(".", 1)[1], n, min(ts), n / min(ts) / 1e6, os.path.getsize("/dev/shm/bench_tag.out.sam") >> 20))
    print("\n".join(l for l in r.stderr.splitlines() if "timing" in l))
for p in (fa, fa + ".fai", sam, bam, "/dev/shm/bench_tag.out.sam"):
    os.remove(p)
