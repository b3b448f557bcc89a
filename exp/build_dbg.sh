#!/bin/bash
# debug build of the C-ABI layer (compiler dumps + scheduler ablation hooks) next to the shipped one: exp/_dbg/libh2e_dbg.so
# (the engine objects are shared).  Use through H2E_LIB=exp/_dbg/libh2e_dbg.so.
set -e
cd "$(dirname "$0")/.."
mkdir -p exp/_dbg
python -m halo2ecc_s_amd.build
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_DEBUG_HOOKS -c halo2ecc_s_amd/csrc/h2e_capi.cpp -o exp/_dbg/h2e_capi_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_dbg.so halo2ecc_s_amd/csrc/engine_fp0.o halo2ecc_s_amd/csrc/engine_fp1.o halo2ecc_s_amd/csrc/engine_fp2.o exp/_dbg/h2e_capi_dbg.o
ls -la exp/_dbg/libh2e_dbg.so
