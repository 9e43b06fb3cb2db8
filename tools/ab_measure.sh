#!/bin/bash
# A/B of one measure's kernels on one box: saved builds (ab/*.so) against the tree's library, config 2 and config-3 density.
# usage: tools/ab_measure.sh pairs|quartet|mhl|fdrp [ab/libA.so ...] ; writes gpurun_out/ab_<measure>.log
what=$1; shift
mkdir -p gpurun_out
out=gpurun_out/ab_$what.log
: > $out
for rnd in 1 2; do
  for lib in "$@" tree; do
    echo "== $lib" >> $out
    if [ $lib = tree ]; then unset METHEOR_HIP_LIB; else export METHEOR_HIP_LIB=$PWD/$lib; fi
    python tools/run_measure_loop.py $what 2>&1 | tail -1 >> $out
    sp=$what; [ $what = quartet ] && sp=me/pm; [ $what = fdrp ] && sp=fdrp+
    python tools/time_sparse.py --only $sp 2>&1 | tail -1 >> $out
  done
done
cat $out
