#!/bin/bash
# Host-side sanitizer pass over the C-ABI layer (program recording + compiler; no GPU needed): builds h2e_capi.cpp with
# -fsanitize=address,undefined (device code unsanitized), links it with the engine objects and runs the CPU tests that
# record programs through it (the product has no library switch: the sanitized build is swapped in under halo2ecc_s_amd/libh2e.so for
# the run and the shipped one put back).  GPU AddressSanitizer is not available on the pool.
set -e
cd "$(dirname "$0")/.."
mkdir -p exp/_dbg
python -m halo2ecc_s_amd.build
C=halo2ecc_s_amd/csrc
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer \
    -c $C/h2e_capi.cpp -o exp/_dbg/h2e_capi_asan.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o exp/_dbg/libh2e_asan.so \
    $C/engine_fp0.o $C/engine_fp1.o $C/engine_fp2.o exp/_dbg/h2e_capi_asan.o $C/checker.o $C/handoff.o
ASAN_LIB=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
set +e
LD_PRELOAD=$ASAN_LIB ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    bash exp/with_lib.sh exp/_dbg/libh2e_asan.so -- python -m pytest tests/test_shape_cpu.py -x -q -m "not gpu" "$@"
exit $?
