#!/bin/bash
# diagnostic build: the bn256 (default) or bls12_381 (FPK=1) engine unit with s_memtime stamps in the pairings' value-chain kernels
# (cycles per round kind, per stage of a linear combination, per computing wave, per record-count bucket) -> exp/_dbg/libh2e_stamps.so
set -e
cd "$(dirname "$0")/.."
mkdir -p exp/_dbg
D="${H2E_STAMP_DEFS:-}"
K="${FPK:-0}"
C=halo2ecc_s_amd/csrc
# The product's engine.hip / wide_int.h carry no timing experiments: the knobs that compute wrong results on purpose (H2E_EXP_MUL_STEPS,
# H2E_EXP_NO_OPS, H2E_EXP_LIN_TERMS, H2E_EXP_NO_HINT_STORES), the A/B forms (H2E_EXP_FERMAT_DIV, H2E_PLAIN_CARRY, H2E_COMPILER_MUL64,
# H2E_EXPERIMENT_NO_INV) and the s_memtime stamps (H2E_WAVE_STAMPS) live in exp/engine_experiments.patch, applied to a scratch copy here.
SRC=exp/_dbg/src; rm -rf $SRC; mkdir -p $SRC; cp halo2ecc_s_amd/csrc/*.h halo2ecc_s_amd/csrc/*.hpp halo2ecc_s_amd/csrc/engine.hip $SRC/
patch -s -d $SRC -p3 < exp/engine_experiments.patch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_FP_ONLY=$K -DH2E_WAVE_STAMPS $D -c $SRC/engine.hip -o exp/_dbg/engine_fp${K}_stamps.o
OBJS=""
for k in 0 1 2; do if [ $k = $K ]; then OBJS="$OBJS exp/_dbg/engine_fp${K}_stamps.o"; else OBJS="$OBJS $C/engine_fp$k.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_stamps.so $OBJS $C/h2e_capi.o $C/checker.o $C/handoff.o
ls -la exp/_dbg/libh2e_stamps.so
