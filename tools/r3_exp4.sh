#!/bin/bash
{
MTH_STREAM=1 timeout 600 python -m pytest tests/test_gpu_pdr_lpmd.py -x -q 2>&1 | tail -2
for o in both lpmd pdr; do echo "== stream only=$o"; ONLY=$o MTH_STREAM=1 python tools/time_tile.py 100 | tail -1; done
echo "== cfg3 stream"; MTH_STREAM=1 python tools/time_sparse.py --only pdr | tail -1
} 2>&1 | grep -v amdgpu.ids
