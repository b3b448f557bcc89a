#!/bin/bash
# A/B of library builds: exp/ab_lib.sh <lib or "default"> ...   (pipelined bench, 2 runs each)
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(round(d["ms_per_step"],2), "chain", [round(x,1) for x in r["value_chain_ms"] if x>0.3], "x", [round(x,1) for x in r["expansion_ms"] if x>0.5], round(r["frac"],3))'
for lib in "$@"; do
  for i in 1 2; do
    echo -n "$lib: "
    if [ "$lib" = default ]; then unset H2E_LIB; else export H2E_LIB=$PWD/$lib; fi
    python bench.py --steps 12 --warmup 4 --no-cpu-baseline --traffic off ${EXTRA} 2>/dev/null | python -c "$show"
  done
done
