#!/bin/bash
# PMC counters of the tile kernel (both forms) on BASELINE config 2; separate --pmc passes, kernel-trace only.
# usage (GPU box, repo root): bash tools/pmc_tile.sh <outdir>
out=${1:-gpurun_out/pmc}; mkdir -p $out
export TMPDIR=/tmp
run() {  # $1 tag, $2 env, rest counters
  tag=$1; envv=$2; shift 2
  env $envv rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$tag -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --soak-seconds 0 --roofline-steps 2 > $out/$tag.log 2>&1
}
for form in calls reads; do
  run ${form}_a METHEOR_TILE_KERNEL=$form SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
  run ${form}_b METHEOR_TILE_KERNEL=$form SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
done
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_[ab]")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "pdr_lpmd" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(d.split("/")[-1], k, {c: round(v / max(n[(k, c)], 1)) for c, v in acc[k].items()})
PY
