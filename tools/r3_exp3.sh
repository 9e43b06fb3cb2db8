#!/bin/bash
{
echo "== cfg3 tile W=4096"; python tools/time_sparse.py --only pdr | tail -1
echo "== cfg3 tile W=8192"; MTH_TILE_W=8192 python tools/time_sparse.py --only pdr | tail -1
echo "== cfg2 tile W=8192"; MTH_TILE_W=8192 python tools/time_tile.py 100 | tail -1
echo "== parity W=8192"; MTH_TILE_W=8192 timeout 600 python -m pytest tests/test_gpu_pdr_lpmd.py -x -q 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids
