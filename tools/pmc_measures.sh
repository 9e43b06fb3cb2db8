#!/bin/bash
# PMC counters of one measure's kernels on config 2; separate --pmc passes, kernel-trace only.  usage: bash tools/pmc_measures.sh <outdir> pairs|mhl|quartet|fdrp
out=${1:-gpurun_out/pmc_m}; what=${2:-pairs}; mkdir -p $out
export TMPDIR=/tmp
run() { tag=$1; shift 1; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$tag -- python tools/run_measure_loop.py $what 6 > $out/$tag.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run c GRBM_GUI_ACTIVE
tail -1 $out/a.log
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/[abc]")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:44]
            if "mth::" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        v = {c: round(x / max(n[(k, c)], 1)) for c, x in acc[k].items()}
        if max(v.values()) > 100000: print(d.split("/")[-1], k, v)
PY
