#!/bin/bash
# all7 leg of bench.py under two settings of one environment switch: bash tools/ab_all7.sh VAR [a] [b]   (default 0 / 1)
v=$1; a=${2:-0}; b=${3:-1}
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-traffic --soak-seconds 0 --preheat-seconds 0"
for rep in 1 2; do
  for x in $a $b; do
    env $v=$x $B 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); a=j['all7']
print('$v=$x', a['per_pass_ms_one_sync_each'], 'seven', a['seven_measures_ms'], 'concurrent', a.get('seven_measures_concurrent_ms'))"
  done
done
