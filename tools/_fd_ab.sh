#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_fdrp.py tests/test_gpu_groups.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
echo "== new"; python tools/time_sparse.py --only fdrp 2>&1 | grep -o 'k_fdrp_walk4": [0-9.]*\|"ms_per_pass": [0-9.]*' | paste - -
done
