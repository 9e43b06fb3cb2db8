#!/bin/bash
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_fdrp.py tests/test_gpu_groups.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2
METHEOR_FDRP_WIN=32 timeout 900 python -m pytest tests/test_gpu_fdrp.py tests/test_gpu_groups.py -x -q 2>&1 | tail -2
for rep in 1 2 3; do
echo "== win32"; METHEOR_FDRP_WIN=32 python tools/time_sparse.py --only fdrp 2>&1 | tail -1
echo "== win64"; METHEOR_FDRP_WIN=64 python tools/time_sparse.py --only fdrp 2>&1 | tail -1
done
