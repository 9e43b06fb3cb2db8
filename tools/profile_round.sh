#!/bin/bash
# Round evidence in one GPU call (repo root, MI355X box): rocprofv3 kernel stats of the bench command, HBM traffic of the
# dominant kernel from two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md section HBM), the bench line.
# usage: bash tools/profile_round.sh <tag> <commit>
tag=${1:-rXX}; commit=${2:-unknown}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-wgbs --no-traffic --soak-seconds 0"
# Per-kernel durations and counters need launches that do not overlap: the engine's batch pipeline (round 4) runs one batch's
# kernels beside the next one's, and a profiler then charges each kernel the time it shared the chip.  The three profiled
# passes switch it off (MTH_PIPELINE=0: the same kernels, one after the other on one stream -- what bench.py's own roofline leg
# times with HIP events); the bench line below is taken with the pipeline on, as shipped.
export MTH_PIPELINE=0
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.log 2>&1
python - "$out" "$commit" <<'PY'
import csv, glob, json, sys, collections
out, commit = sys.argv[1], sys.argv[2]
def counter(d, name):
    acc = collections.defaultdict(list)
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"].split("(")[0].split("<")[0].replace("void mth::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
fetch, write = counter("fetch", "FETCH_SIZE"), counter("write", "WRITE_SIZE")
stats = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0].split("<")[0].replace("void mth::", "")] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3)
    import shutil; shutil.copy(f, out + "/kernel_stats.csv")
dom = max((k for k in stats if "pdr_lpmd" in k), key=lambda k: stats[k][1])
# gfx950: FETCH_SIZE (KB) reports half of a wide coalesced streaming read -- doubled, as MI355X_MICROARCH.md prescribes; WRITE_SIZE as reported
j = {"kernel": dom, "commit": commit, "reads_per_gpu": 10000000, "launches_averaged": stats[dom][0], "rocprof_avg_us": round(stats[dom][1], 2),
     "FETCH_SIZE_KB": round(fetch.get(dom, 0), 1), "WRITE_SIZE_KB": round(write.get(dom, 0), 1),
     "hbm_bytes_per_launch": int(fetch.get(dom, 0) * 1024 * 2 + write.get(dom, 0) * 1024),
     "note": "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md section HBM); separate --pmc passes; command: bench.py --steps 20 --warmup 3"}
json.dump(j, open(out + "/pmc_traffic.json", "w"), indent=1)
print(json.dumps(j))
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1])[:8]: print("%-28s calls %4d avg %.1f us" % (k, v[0], v[1]))
PY
unset MTH_PIPELINE
python bench.py > $out/bench.json 2> $out/bench.err
cut -c1-400 $out/bench.json
