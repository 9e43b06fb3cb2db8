#!/bin/bash
# round 3: streaming PDR+LPMD kernel against the tile pipeline on one box (parity suites first, then config 2 and config-3 density)
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_pdr_lpmd.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15
echo "== config 2, stream"; python tools/time_tile.py 200 2>&1 | tail -1
echo "== config 2, tile";  MTH_NO_STREAM=1 python tools/time_tile.py 200 2>&1 | tail -1
echo "== config 2, stream"; python tools/time_tile.py 200 2>&1 | tail -1
echo "== config 2, tile";  MTH_NO_STREAM=1 python tools/time_tile.py 200 2>&1 | tail -1
echo "== config 3 density chr1, stream"; python tools/time_sparse.py --only pdr 2>&1 | tail -2
echo "== config 3 density chr1, tile"; MTH_NO_STREAM=1 python tools/time_sparse.py --only pdr 2>&1 | tail -2
} > gpurun_out/r3_stream_ab.log 2>&1
cat gpurun_out/r3_stream_ab.log
