#!/bin/bash
# A/B of the PDR+LPMD tile kernel on one box: the saved build (ab/libmetheor_hip_r02base.so) against the tree's variants.
# usage: tools/ab_tile.sh [reps] ; writes gpurun_out/ab_tile.log
mkdir -p gpurun_out
out=gpurun_out/ab_tile.log
: > $out
for rnd in 1 2; do
  if [ -f ab/libmetheor_hip_r02base.so ]; then
    echo "== base (round-2 start)" >> $out
    METHEOR_HIP_LIB=$PWD/ab/libmetheor_hip_r02base.so python tools/tile_tail_probe.py 8 ${1:-40} 2>&1 | tail -6 >> $out
  fi
  for v in 0 1; do
    echo "== MTH_TILE_VARIANT=$v" >> $out
    MTH_TILE_VARIANT=$v python tools/tile_tail_probe.py 8 ${1:-40} 2>&1 | tail -6 >> $out
  done
done
