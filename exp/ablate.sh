#!/bin/bash
# What does each kernel class cost the pipelined MSM step?  Debug build (exp/build_dbg.sh) with H2E_DEBUG_SKIP masks: the arrays
# of such runs are garbage, only the step time means something.  Run on the GPU box from the repo root.
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(round(d["ms_per_step"],2), "chain", [round(x,1) for x in r["value_chain_ms"] if x>0.3], "x", [round(x,1) for x in r["expansion_ms"] if x>0.5])'
export H2E_LIB=$PWD/exp/_dbg/libh2e_dbg.so
for skip in ${SKIPS:-0 1 2 4 8 16 32 64 3 7 47 63 127}; do
  echo -n "skip $skip: "
  H2E_DEBUG_SKIP=$skip python bench.py --steps 12 --warmup 4 --no-cpu-baseline --traffic off --no-check 2>/dev/null | python -c "$show"
done
