#!/bin/bash
# interleaved A/B of library builds on one box, FDRP pass on a chr1-sized contig: tools/ab_wtile.sh <rounds> abx/libA.so ... (the tree's build last)
rounds=${1:-3}; shift
for r in $(seq $rounds); do
  for lib in "$@"; do echo "== $lib: $(METHEOR_FDRP_WTILE=1 METHEOR_HIP_LIB=$PWD/$lib python tools/time_sparse.py --only fdrp 2>&1 | tail -1 | cut -c1-400)"; done
  echo "== tree: $(METHEOR_FDRP_WTILE=1 python tools/time_sparse.py --only fdrp 2>&1 | tail -1 | cut -c1-400)"
done
