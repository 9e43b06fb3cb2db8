#!/bin/bash
# timeline of one `metheor pdr` run on a config-2 BAM: kernels and memory copies longer than 2 ms (rocprofv3 traces)
export TMPDIR=/tmp
python - <<PY
import sys
sys.path.insert(0, ".")
from metheor_amd import hostapi, synth
hostapi.write_synthetic_bam("/dev/shm/t.bam", synth.chr19_10m(), contig="chr19", seed=7)
PY
./metheor_amd/metheor pdr -i /dev/shm/t.bam -o /dev/shm/t.tsv
METHEOR_TEARDOWN=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r02_e2e_trace -- ./metheor_amd/metheor pdr -i /dev/shm/t.bam -o /dev/shm/t.tsv > /dev/null 2>&1
python - <<PY
import csv, glob
rows = []
for f in glob.glob("gpurun_out/r02_e2e_trace/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48]) for r in csv.DictReader(open(f))]
for f in glob.glob("gpurun_out/r02_e2e_trace/**/*memory_copy_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r["Direction"]) for r in csv.DictReader(open(f))]
rows.sort()
if rows:
    t0 = rows[0][0]
    for a, b, n in rows:
        if b - a > 300_000: print("%8.1f ms  +%7.1f ms  %s" % ((a - t0) / 1e6, (b - a) / 1e6, n))
    print("span ms %.1f, %d events" % ((rows[-1][1] - t0) / 1e6, len(rows)))
else:
    print("no events")
PY
rm -f /dev/shm/t.bam /dev/shm/t.tsv
