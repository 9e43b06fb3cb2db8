#!/bin/bash
# interleaved A/B of saved library builds (ab/*.so) against the tree's on one box: tools/ab_lib.sh "<command>" [rounds] ab/libA.so ...
cmd=$1; rounds=${2:-3}; shift 2
for r in $(seq $rounds); do
  for lib in "$@"; do echo "== $lib: $(METHEOR_HIP_LIB=$PWD/$lib $cmd 2>&1 | tail -1)"; done
  echo "== tree: $($cmd 2>&1 | tail -1)"
done
