#!/bin/bash
{
for c in 128 256 384 512 640 1280 2560; do for o in both lpmd; do echo "== stream C=$c only=$o"; MTH_STREAM_C=$c ONLY=$o MTH_STREAM=1 python tools/time_tile.py 100 | tail -1; done; done
} 2>&1 | grep -v amdgpu.ids
