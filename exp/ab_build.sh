#!/bin/bash
# A/B build of one engine unit with extra defines -> exp/_dbg/libh2e_<name>.so: exp/ab_build.sh <name> <unit 0|1|2> <defines...>
# (use: copy it over halo2ecc_s_amd/libh2e.so in the GPU box's scratch copy of the tree, run, copy the shipped one back)
set -e
cd "$(dirname "$0")/.."
mkdir -p exp/_dbg
N=$1; K=$2; shift; shift
C=halo2ecc_s_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_FP_ONLY=$K "$@" -c $C/engine.hip -o exp/_dbg/engine_fp${K}_$N.o
OBJS=""
for k in 0 1 2; do if [ $k = $K ]; then OBJS="$OBJS exp/_dbg/engine_fp${K}_$N.o"; else OBJS="$OBJS $C/engine_fp$k.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_$N.so $OBJS $C/h2e_capi.o $C/checker.o $C/handoff.o
ls -la exp/_dbg/libh2e_$N.so
