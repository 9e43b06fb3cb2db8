#!/bin/bash
# split form of the tile kernel (MTH_TILE_NA = 0 / 2 / 4): parity, then config 2 and config-3 density
{
for na in 2 4; do echo "== parity NA=$na"; MTH_TILE_NA=$na timeout 900 python -m pytest tests/test_gpu_pdr_lpmd.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -3; done
for na in 0 4 2 0 4; do echo "== cfg2 NA=$na"; MTH_TILE_NA=$na python tools/time_tile.py 200 | tail -1; done
for na in 0 2 4 0 2; do echo "== cfg3 NA=$na"; MTH_TILE_NA=$na python tools/time_sparse.py --only pdr | tail -1; done
} 2>&1 | grep -v amdgpu.ids
