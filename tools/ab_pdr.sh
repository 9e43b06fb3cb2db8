#!/bin/bash
# interleaved A/B of library builds on one box, PDR + LPMD pass on a chr1-sized contig: tools/ab_pdr.sh <rounds> abx/libA.so ... (the tree's build last)
rounds=${1:-3}; shift
for r in $(seq $rounds); do
  for lib in "$@"; do echo "== $lib: $(METHEOR_HIP_LIB=$PWD/$lib python tools/time_sparse.py --only pdr 2>&1 | tail -1 | cut -c1-400)"; done
  echo "== tree: $(python tools/time_sparse.py --only pdr 2>&1 | tail -1 | cut -c1-400)"
done
