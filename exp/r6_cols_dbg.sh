#!/bin/bash
# where the column-emission expansion's time goes (wrong results on purpose: --no-check).  The knobs are NOT in the product: this builds
# the bn256 column unit from a scratch copy with exp/engine_experiments.patch + the debug build of the C-ABI layer (which reads
# H2E_COLS_DBG) into exp/_dbg/libh2e_colsdbg.so and runs the bench's consumer-ready child under exp/with_lib.sh.
#   H2E_COLS_DBG bits: 1 = no working-copy stores, 2 = no column stores, 4 = no zero fill of passed rows, 8 = no hint loads
O=${1:-gpurun_out/r6_cols}; mkdir -p $O exp/_dbg
C=halo2ecc_s_amd/csrc
SRC=exp/_dbg/src; rm -rf $SRC; mkdir -p $SRC; cp $C/*.h $C/*.hpp $C/engine.hip $SRC/
patch -s -d $SRC -p3 < exp/engine_experiments.patch || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_FP_ONLY=0 -DH2E_COLS -c $SRC/engine.hip -o exp/_dbg/engine_cols_fp0_dbg.o || exit 1
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DH2E_DEBUG_HOOKS -c $C/h2e_capi.cpp -o exp/_dbg/h2e_capi_dbg.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/_dbg/libh2e_colsdbg.so $C/engine_fp0.o $C/engine_fp1.o $C/engine_fp2.o exp/_dbg/h2e_capi_dbg.o \
    exp/_dbg/engine_cols_fp0_dbg.o $C/engine_cols_fp1.o $C/engine_cols_fp2.o $C/checker.o $C/handoff.o || exit 1
B="--sub --suite main --workload msm --ring 1 --steps 2 --warmup 1 --latency-steps 0 --consumer-ready 2 --no-cpu-baseline --traffic off --full-line --no-check"
for d in ${DBGS:-0 1 2 3 4 7}; do
  H2E_COLS_DBG=$d bash exp/with_lib.sh exp/_dbg/libh2e_colsdbg.so -- python bench.py $B > $O/dbg_$d.json 2> $O/dbg_$d.err
  python - $O/dbg_$d.json $d <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1])
fp=d["consumer_ready_first_pass"]
lm=fp.get("launch_ms",[[]])[-1]
print("dbg", sys.argv[2], "first pass ms", fp.get("ms_per_step"), "windows", lm[8] if len(lm)>8 else None, "cand", lm[5] if len(lm)>5 else None, fp.get("error"))
P
done
