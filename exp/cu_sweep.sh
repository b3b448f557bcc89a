#!/bin/bash
# H2E_CU_RESERVE sweep: CUs set aside for the value-chain streams of pipelined runs (run on the GPU box from the repo root)
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(round(d["ms_per_step"],2), "chain", [round(x,1) for x in r["value_chain_ms"] if x>0.3], "x", [round(x,1) for x in r["expansion_ms"] if x>0.5])'
for cfg in ${CFGS:-"0" "64,0,0" "64,1,0" "96,0,0" "96,1,0" "128,1,0" "0"}; do
  echo -n "cu $cfg: "
  H2E_CU_RESERVE=$cfg python bench.py --steps 12 --warmup 4 --no-cpu-baseline --traffic off ${EXTRA} 2>/dev/null | python -c "$show"
done
