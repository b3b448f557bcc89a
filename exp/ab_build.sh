#!/bin/bash
# A/B build of the bn256 engine unit with extra defines: exp/ab_build.sh <name> <defines...>  ->  exp/_dbg/libh2e_<name>.so  (use through H2E_LIB=...)
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p exp/_dbg
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_FP_ONLY=0 "$@" -c halo2ecc_s_amd/csrc/engine.hip -o exp/_dbg/engine_fp0_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_$N.so exp/_dbg/engine_fp0_$N.o halo2ecc_s_amd/csrc/engine_fp1.o halo2ecc_s_amd/csrc/engine_fp2.o halo2ecc_s_amd/csrc/h2e_capi.o
