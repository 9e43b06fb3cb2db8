#!/bin/bash
# PMC counters of k_inflate on a 2 M-read synthetic BAM; separate --pmc passes, kernel-trace only.
# usage (GPU box, repo root): bash tools/pmc_inflate.sh <outdir>
out=${1:-gpurun_out/pmc_inflate}; mkdir -p $out
export TMPDIR=/tmp
run() { tag=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$tag -- python tools/time_inflate.py 2000000 > $out/$tag.log 2>&1; }
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES
run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
python - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/[ab]")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40]
            if "inflate" not in k and "crc32" not in k and "k_decode" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        print(d.split("/")[-1], k, {c: round(v / max(n[(k, c)], 1)) for c, v in acc[k].items()})
PY
