#!/bin/bash
# A/B of the PDR+LPMD tile kernel on one box: saved builds (ab/*.so given as arguments) against the tree's library.
# usage: tools/ab_tile.sh [reps] [ab/libA.so ...] ; writes gpurun_out/ab_tile.log
reps=${1:-40}; shift
mkdir -p gpurun_out
out=gpurun_out/ab_tile.log
: > $out
for rnd in 1 2; do
  for lib in "$@"; do
    echo "== $lib" >> $out
    METHEOR_HIP_LIB=$PWD/$lib python tools/tile_tail_probe.py 8 $reps 2>&1 | tail -6 >> $out
  done
  echo "== tree" >> $out
  python tools/tile_tail_probe.py 8 $reps 2>&1 | tail -6 >> $out
done
