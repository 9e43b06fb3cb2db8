#!/bin/bash
# PMC counters of k_mhl_tile (+ k_mhl_rowcheck) on one chr1-sized contig at config-3 density (tools/time_sparse.py --only mhl); separate --pmc
# passes, kernel-trace only.  usage (GPU box, repo root): bash tools/pmc_mhl.sh <outdir>
out=${1:-gpurun_out/pmc_mhl}; mkdir -p $out
export TMPDIR=/tmp

run() {  # $1 tag, rest counters
  tag=$1; shift 1
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$tag -- python tools/time_sparse.py --only mhl > $out/$tag.log 2>&1
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
run c GRBM_GUI_ACTIVE
run d SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/[abcd]")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "mhl" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(d.split("/")[-1], k, {c: round(v / max(n[(k, c)], 1)) for c, v in acc[k].items()})
PY
