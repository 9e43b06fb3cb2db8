"""One `metheor` run split over the GPUs of a node -- a thin launcher kept for scripts written against round 1.

    python -m metheor_amd.sharded --shards 8 -- pdr -i in.bam -o out.tsv [flags]

is exactly `metheor pdr -i in.bam -o out.tsv [flags] --gpus 8`: the executable itself runs one host thread and one device
context per shard (shard r on device r mod the devices in use), every shard plans and loads its own run of BGZF blocks
(mth_host_plan_shard: no index, no router), owns a (tid, pos) interval, and the parts are concatenated in memory in shard
order; `lpmd` sums its four global counters with one RCCL all-reduce (mth_allreduce_lpmd).  No temporary files, nothing
parsed here (cli_main.cpp).
"""
import argparse
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def run(n_shards, metheor_args, gpus=None, exe=None, env=None):
    """gpus: devices to place the shards on (default: all present); returns the executable's exit status"""
    exe = exe or os.path.join(_HERE, "metheor")
    e = dict(os.environ if env is None else env)
    if gpus:
        e["METHEOR_DEVICES"] = str(gpus)
    p = subprocess.run([exe] + list(metheor_args) + ["--gpus", str(n_shards)], env=e, stderr=subprocess.PIPE, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stderr)
    return p.returncode


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--shards", type=int, required=True, help="number of shards (= metheor --gpus)")
    ap.add_argument("--gpus", type=int, default=None, help="devices to place the shards on (default: all present)")
    ap.add_argument("rest", nargs=argparse.REMAINDER, help="-- <metheor arguments>")
    a = ap.parse_args(argv)
    rest = a.rest[1:] if a.rest and a.rest[0] == "--" else a.rest
    if not rest:
        ap.error("missing the metheor command line after --")
    return run(a.shards, rest, a.gpus)


if __name__ == "__main__":
    sys.exit(main())
