#!/bin/bash
# debug build of the C-ABI layer (compiler dumps, scheduler ablation hooks, H2E_DEBUG_LOG launch log) next to the shipped one:
# exp/_dbg/libh2e_dbg.so (the engine objects are shared).  The product has no library switch: a script that wants it copies it over
# halo2ecc_s_amd/libh2e.so in the GPU box's scratch copy of the tree, or patches halo2ecc_s_amd.engine.lib_path (exp/wave_stamps.py).
set -e
cd "$(dirname "$0")/.."
mkdir -p exp/_dbg
C=halo2ecc_s_amd/csrc
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_DEBUG_HOOKS -c $C/h2e_capi.cpp -o exp/_dbg/h2e_capi_dbg.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_dbg.so $C/engine_fp0.o $C/engine_fp1.o $C/engine_fp2.o exp/_dbg/h2e_capi_dbg.o $C/checker.o $C/handoff.o
ls -la exp/_dbg/libh2e_dbg.so
