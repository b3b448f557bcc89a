#!/bin/bash
# diagnostic build: the bn256 engine unit with s_memtime stamps in h2e_replay_wave (cycles per round kind), as exp/_dbg/libh2e_stamps.so
set -e
cd "$(dirname "$0")/.."
D="${H2E_STAMP_DEFS:-}"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_FP_ONLY=0 -DH2E_WAVE_STAMPS $D -c halo2ecc_s_amd/csrc/engine.hip -o exp/_dbg/engine_fp0_stamps.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_stamps.so exp/_dbg/engine_fp0_stamps.o halo2ecc_s_amd/csrc/engine_fp1.o halo2ecc_s_amd/csrc/engine_fp2.o halo2ecc_s_amd/csrc/h2e_capi.o
ls -la exp/_dbg/libh2e_stamps.so
