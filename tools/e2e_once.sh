#!/bin/bash
# itemised timing lines of `metheor pdr` on a config-2 BAM (third of three runs)
python - <<PY
import sys
sys.path.insert(0, ".")
from metheor_amd import hostapi, synth
hostapi.write_synthetic_bam("/dev/shm/t.bam", synth.chr19_10m(), contig="chr19", seed=7)
PY
for k in 1 2 3 4 5; do
  s=$(date +%s.%N); METHEOR_TIMING=1 ./metheor_amd/metheor pdr -i /dev/shm/t.bam -o /dev/shm/t.tsv 2> /tmp/e.$k; e=$(date +%s.%N); python3 -c "print('wall %.3f s' % ($e - $s))" >> /tmp/e.$k
done
cat /tmp/e.3; echo ----; cat /tmp/e.5
rm -f /dev/shm/t.bam /dev/shm/t.tsv
