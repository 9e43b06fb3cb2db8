"""Region sharding of one contig's SoA: per-site outputs of disjoint regions are
independent, so ranks / batches own disjoint [beg,end) windows and only re-read a halo of reads.

Host-side index arithmetic only (numpy); the same slices are what a streaming C++ host emits."""
import numpy as np


def max_span(c):
    if len(c["read_start"]) == 0:
        return 0
    return int((c["read_end"].astype(np.int64) - c["read_start"].astype(np.int64)).max()) + 1


def plan_regions(c, n_parts):
    """split [0, length) into n_parts windows holding about equal numbers of read starts"""
    n = len(c["read_start"])
    cuts = [0]
    for k in range(1, n_parts):
        cuts.append(int(c["read_start"][min(n - 1, (n * k) // n_parts)]) if n else 0)
    cuts.append(int(c["length"]))
    cuts = np.maximum.accumulate(np.array(cuts, dtype=np.int64))
    return [(int(cuts[k]), int(cuts[k + 1])) for k in range(n_parts)]


def slice_region(c, beg, end, halo=None):
    """reads that can touch sites in [beg,end): start in [beg-halo, end]; CSR rebased.
    A call sits in [start-1, end_of_read], so halo = max_span covers the left side and a reverse
    read starting exactly at `end` still reports end-1."""
    if halo is None:
        halo = max_span(c)
    s = c["read_start"]
    i0 = int(np.searchsorted(s, beg - halo, side="left"))
    i1 = int(np.searchsorted(s, end, side="right"))
    o0, o1 = int(c["cpg_off"][i0]), int(c["cpg_off"][i1])
    out = dict(c)
    for k in ("read_start", "read_end", "read_mapq", "read_fwd"):
        out[k] = c[k][i0:i1]
    out["cpg_off"] = (c["cpg_off"][i0:i1 + 1].astype(np.int64) - o0).astype(np.uint32)
    out["cpg_pos"] = c["cpg_pos"][o0:o1]
    out["cpg_rel"] = c["cpg_rel"][o0:o1]
    out["region"] = (beg, end)
    return out


def plan_genome(contig_reads, world):
    """Region sharding of one genome over `world` ranks: the reads in genome order are cut into `world` runs of equal
    length, so a rank owns a contiguous stretch of the genome -- whole contigs and, where a cut falls inside one, a region of it.
    contig_reads: reads per contig, in tid order.  Returns, per rank, a list of (tid, first_read, end_read) with contig-local read
    indices; the caller turns the indices into positions (region = [start[first_read] or 0, start[end_read] or contig length)),
    which is what makes ownership a property of the POSITION (a read is owned by the rank whose region holds its start)."""
    total = int(sum(contig_reads))
    cuts = [(total * r) // world for r in range(world + 1)]
    plan = [[] for _ in range(world)]
    base = 0
    for tid, n in enumerate(contig_reads):
        n = int(n)
        for r in range(world):
            a, b = max(cuts[r], base), min(cuts[r + 1], base + n)
            if b > a:
                plan[r].append((tid, a - base, b - base))
        base += n
    return plan
