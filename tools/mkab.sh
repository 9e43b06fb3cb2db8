#!/bin/bash
# build an A/B variant of libmetheor_hip.so from the tree's objects with mth_pdr_lpmd.hip recompiled under extra flags:
# tools/mkab.sh <name> <flags...>  ->  abx/lib<name>.so   (abx/ travels to the GPU box; delete it when the A/B is over)
name=$1; shift
src=${MKAB_SRC:-mth_pdr_lpmd}      # which source is recompiled under the extra flags
cd metheor_amd/csrc || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -ffp-contract=off -Wno-unused-result "$@" -c $src.hip -o /tmp/ab_$name.o 2>&1 | grep -E "error" 
objs=$(ls *.o | grep -v $src.o | grep -v mth_sites_wide.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../abx/lib$name.so /tmp/ab_$name.o $objs -ldl && echo built abx/lib$name.so
