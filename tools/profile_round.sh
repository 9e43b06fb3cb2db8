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
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B > $out/stats.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.log 2>&1
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
# ---- the other legs that carry a reported number: rocprofv3 kernel stats of `bench.py --legs all7` (config 3: every
# measure over the two contig groups) and `--legs fdrp_pairs` (config 4), and ONE PMC pass each -- vector-unit busy, waits, LDS conflicts,
# bytes fetched / written -- for k_fdrp_tile / k_fdrp_chain / k_fdrp_walk / k_fdrp_walk4 / k_mhl_tile / k_mhl_walk_big / k_quartet_tile /
# k_pairs_tile / k_pdr_lpmd_wide.  (Queued ME / PM / pairs batches and pipelined PDR batches overlap kernels of neighbouring batches: the
# profiled runs switch both off, as above.)
export MTH_QUARTET_QUEUE=0 MTH_PAIRS_QUEUE=0
for leg in all7 fdrp_pairs; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${leg}_stats -- python bench.py --legs $leg > $out/${leg}.json 2> $out/${leg}.err
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/${leg}_pmc1 -- python bench.py --legs $leg > /dev/null 2> $out/${leg}_pmc1.err
  # (FETCH_SIZE and WRITE_SIZE in passes of their own: together with a third counter the request "exceeds the capabilities of the hardware"
  # and rocprofv3 then hangs in its abort handler -- every profiled command runs under `timeout`)
  timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/${leg}_pmc2 -- python bench.py --legs $leg > /dev/null 2> $out/${leg}_pmc2.err
  timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/${leg}_pmc3 -- python bench.py --legs $leg > /dev/null 2> $out/${leg}_pmc3.err
  timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/${leg}_pmc4 -- python bench.py --legs $leg > /dev/null 2> $out/${leg}_pmc4.err
done
unset MTH_QUARTET_QUEUE MTH_PAIRS_QUEUE
python - "$out" <<'PY'
import csv, glob, json, sys, collections, shutil
out = sys.argv[1]
short = lambda k: k.split("(")[0].replace("void ", "").replace("mth::", "")[:48]
for leg in ("all7", "fdrp_pairs"):
    for f in glob.glob(out + "/%s_stats/**/*kernel_stats.csv" % leg, recursive=True):
        shutil.copy(f, out + "/%s_kernel_stats.csv" % leg)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in ("pmc1", "pmc2", "pmc3", "pmc4"):
        for f in glob.glob(out + "/%s_%s/**/*counter_collection.csv" % (leg, d), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k.startswith("k_"): continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows = {k: {c: sum(v) / len(v) for c, v in cs.items()} | {"launches_profiled": max(len(v) for v in cs.values())} for k, cs in acc.items()}
    # gfx950: FETCH_SIZE is in KB and reports half of a wide streaming read (MI355X_MICROARCH.md, HBM section): doubled here; WRITE_SIZE as reported
    for k, r in rows.items():
        if "FETCH_SIZE" in r: r["hbm_read_MB_per_launch"] = round(r["FETCH_SIZE"] * 2 * 1024 / 1e6, 2)
        if "WRITE_SIZE" in r: r["hbm_write_MB_per_launch"] = round(r["WRITE_SIZE"] * 1024 / 1e6, 2)
        if r.get("SQ_WAVE_CYCLES"): r["wait_any_share"] = round(r.get("SQ_WAIT_ANY", 0) / r["SQ_WAVE_CYCLES"], 3); r["wait_issue_share"] = round(r.get("SQ_WAIT_INST_ANY", 0) / r["SQ_WAVE_CYCLES"], 3)
        if r.get("GRBM_GUI_ACTIVE") and r.get("SQ_ACTIVE_INST_VALU"): r["valu_quad_cycles_per_simd_over_kernel_cycles"] = round(r["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (r["GRBM_GUI_ACTIVE"] / 8), 3)
    json.dump(rows, open(out + "/%s_pmc.json" % leg, "w"), indent=1, sort_keys=True)
    print(leg, "PMC kernels:", sorted(rows))
PY
unset MTH_PIPELINE
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err
cut -c1-400 $out/bench.json
