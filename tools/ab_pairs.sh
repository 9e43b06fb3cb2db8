#!/bin/bash
# A/B of k_pairs_tile on one box: saved builds (ab/*.so given as arguments) against the tree's library, config 2 and config-3 density.
# usage: tools/ab_pairs.sh [ab/libA.so ...] ; writes gpurun_out/ab_pairs.log
mkdir -p gpurun_out
out=gpurun_out/ab_pairs.log
: > $out
for rnd in 1 2; do
  for lib in "$@" tree; do
    echo "== $lib" >> $out
    if [ $lib = tree ]; then unset METHEOR_HIP_LIB; else export METHEOR_HIP_LIB=$PWD/$lib; fi
    python tools/run_measure_loop.py pairs 2>&1 | tail -1 >> $out
    python tools/time_sparse.py --only pairs 2>&1 | tail -1 >> $out
  done
done
cat $out
