"""One `metheor` run as N shards, one process per GPU.

    python -m metheor_amd.sharded --shards 8 -- pdr -i in.bam -o out.tsv [flags]

Every shard is the stand-alone `metheor` executable with METHEOR_SHARD=r/N (and METHEOR_DEVICE = r mod --gpus): it plans
its own run of BGZF blocks (mth_host_plan_shard: no index, no router process), loads and decodes them on its GPU, owns a
(tid, pos) interval and writes <output>.shard-r-of-N.  There is no data-path exchange between shards: per-site, per-quartet
and per-pair rows are owned by position.  The merge concatenates the parts in shard order (= genome order) and, for
`lpmd`, sums the four global counters (the one reduction the path has; bench.py does it with an RCCL all-reduce on
device-resident batches) and applies lpmd.rs:145-147 to the sum.
"""
import argparse
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _outputs(args):
    outs = []
    for k, a in enumerate(args):
        if a in ("-o", "--output", "--pairs") and k + 1 < len(args):
            outs.append(args[k + 1])
        elif a.startswith("--output=") or a.startswith("--pairs="):
            outs.append(a.split("=", 1)[1])
    return outs


def merge(path, n):
    parts = ["%s.shard-%d-of-%d" % (path, r, n) for r in range(n)]
    with open(parts[0], "rb") as f:
        head = f.read(12)
    if head.startswith(b"#lpmd_counts"):
        from . import capi, hostapi
        tot = [0, 0, 0, 0]
        name = ""
        for p in parts:
            f = open(p).read().rstrip("\n").split("\t")
            name = f[1]
            for k in range(4):
                tot[k] += int(f[2 + k])
        # the reference's counters are i32 and wrap in release builds (lpmd.rs:32-41): mth_lpmd_from_counts applies that
        v = capi.lib().mth_lpmd_from_counts(tot[0], tot[1])
        with open(path, "w") as out:
            out.write("name\tlpmd\n%s\t%s\n" % (name, hostapi.format_f32(v)))
    else:
        with open(path, "wb") as out:
            for p in parts:
                with open(p, "rb") as f:
                    shutil.copyfileobj(f, out, 1 << 24)
    for p in parts:
        os.remove(p)


def run(n_shards, metheor_args, gpus=None, exe=None, env=None):
    exe = exe or os.path.join(_HERE, "metheor")
    gpus = gpus or n_shards
    procs = []
    for r in range(n_shards):
        e = dict(os.environ if env is None else env)
        e["METHEOR_SHARD"] = "%d/%d" % (r, n_shards)
        e["METHEOR_DEVICE"] = str(r % gpus)
        procs.append(subprocess.Popen([exe] + list(metheor_args), env=e, stderr=subprocess.PIPE, text=True))
    rc = 0
    for r, p in enumerate(procs):
        _, err = p.communicate()
        if p.returncode != 0:
            rc = rc or p.returncode
            sys.stderr.write("[shard %d] %s" % (r, err))
    if rc:
        return rc
    for o in _outputs(metheor_args):
        merge(o, n_shards)
    return 0


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--shards", type=int, required=True, help="number of shards (one process each)")
    ap.add_argument("--gpus", type=int, default=None, help="GPUs on this node (default: one per shard)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="-- <metheor arguments>")
    a = ap.parse_args(argv)
    rest = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest
    if not rest:
        ap.error("missing the metheor command line after --")
    return run(a.shards, rest, a.gpus)


if __name__ == "__main__":
    sys.exit(main())
