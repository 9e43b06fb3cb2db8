#!/bin/bash
# The wide (hashed-site) PDR+LPMD kernel on a config-3-density contig: rocprofv3 kernel stats, then separate --pmc passes, next to
# the dense tile kernel forced on the same data (MTH_PDR_WIDE=0).  usage (GPU box, repo root): bash tools/pmc_wide.sh <outdir>
out=${1:-gpurun_out/pmc_wide}; mkdir -p $out
export TMPDIR=/tmp
cmd="python tools/time_sparse.py --only pdr"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $cmd > $out/stats.log 2>&1
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
run() { tag=$1; shift 1
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/w$tag -- $cmd > $out/w$tag.log 2>&1
  MTH_PDR_WIDE=0 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/d$tag -- $cmd > $out/d$tag.log 2>&1
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR
run c GRBM_GUI_ACTIVE
run f FETCH_SIZE
run g WRITE_SIZE
head -12 $out/kernel_stats.csv | cut -c1-160
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/[wd][abcfg]")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            if "pdr_lpmd" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(d.split("/")[-1], k, {c: round(v / max(n[(k, c)], 1)) for c, v in acc[k].items()})
PY
