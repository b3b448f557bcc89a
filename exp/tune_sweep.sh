#!/bin/bash
# H2E_TUNE="reserve,xcache,xpad" sweep of the pipelined bench (run on the GPU box from the repo root)
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(round(d["ms_per_step"],2), "chain", [round(x,1) for x in r["value_chain_ms"] if x>0.3], "x", [round(x,1) for x in r["expansion_ms"] if x>0.5])'
for cfg in "143360,1,0" "0,1,0" "0,0,0" "0,0,20480" "0,0,36864" "0,1,6144"; do
  for ring in 2 1; do
    echo -n "tune $cfg ring $ring: "
    H2E_TUNE=$cfg python bench.py --steps 10 --warmup 4 --no-cpu-baseline --traffic off --ring $ring 2>/dev/null | python -c "$show"
  done
done
