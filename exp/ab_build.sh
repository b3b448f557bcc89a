#!/bin/bash
# A/B build of one engine unit with extra defines -> exp/_dbg/libh2e_<name>.so: exp/ab_build.sh <name> <unit 0|1|2> <defines...>
# (use: copy it over halo2ecc_s_amd/libh2e.so in the GPU box's scratch copy of the tree, run, copy the shipped one back)
set -e
cd "$(dirname "$0")/.."
mkdir -p exp/_dbg
N=$1; K=$2; shift; shift
C=halo2ecc_s_amd/csrc
# The product's engine.hip / wide_int.h carry no timing experiments: the knobs that compute wrong results on purpose (H2E_EXP_MUL_STEPS,
# H2E_EXP_NO_OPS, H2E_EXP_LIN_TERMS, H2E_EXP_NO_HINT_STORES), the A/B forms (H2E_EXP_FERMAT_DIV, H2E_PLAIN_CARRY, H2E_COMPILER_MUL64,
# H2E_EXPERIMENT_NO_INV) and the s_memtime stamps (H2E_WAVE_STAMPS) live in exp/engine_experiments.patch, applied to a scratch copy here.
SRC=exp/_dbg/src; rm -rf $SRC; mkdir -p $SRC; cp halo2ecc_s_amd/csrc/*.h halo2ecc_s_amd/csrc/*.hpp halo2ecc_s_amd/csrc/engine.hip $SRC/
patch -s -d $SRC -p3 < exp/engine_experiments.patch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_FP_ONLY=$K "$@" -c $SRC/engine.hip -o exp/_dbg/engine_fp${K}_$N.o
OBJS=""
for k in 0 1 2; do if [ $k = $K ]; then OBJS="$OBJS exp/_dbg/engine_fp${K}_$N.o"; else OBJS="$OBJS $C/engine_fp$k.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_$N.so $OBJS $C/h2e_capi.o $C/checker.o $C/handoff.o
ls -la exp/_dbg/libh2e_$N.so
