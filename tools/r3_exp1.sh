#!/bin/bash
# experiment: stream kernel by halves, and at 4 waves per SIMD (LDS padded)
{
for o in both pdr lpmd; do echo "== stream only=$o"; ONLY=$o MTH_STREAM=1 python tools/time_tile.py 100 | tail -1; echo "== tile only=$o"; ONLY=$o python tools/time_tile.py 100 | tail -1; done
echo "== stream, padded LDS (4 WGs per CU)"; METHEOR_HIP_LIB=$PWD/ab/libpad.so MTH_STREAM=1 python tools/time_tile.py 100 | tail -1
echo "== stream, padded LDS (6 WGs per CU)"; METHEOR_HIP_LIB=$PWD/ab/libpad6.so MTH_STREAM=1 python tools/time_tile.py 100 | tail -1
} 2>&1 | grep -v amdgpu.ids
