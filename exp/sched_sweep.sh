#!/bin/bash
# H2E_SCHED / H2E_SMALL_X_LANES sweep (which stream a pipelined run's small expansions / fix-ups use); run on the GPU box from the repo root
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(round(d["ms_per_step"],2), "chain", [round(x,1) for x in r["value_chain_ms"] if x>0.3], "x", [round(x,1) for x in r["expansion_ms"] if x>0.5])'
for cfg in "4 262144" "12 262144" "8 262144" "4 262144" "12 262144"; do
  set -- $cfg
  echo -n "sched $1 small_x_lanes $2: "
  H2E_SCHED=$1 H2E_SMALL_X_LANES=$2 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --traffic off 2>/dev/null | python -c "$show"
done
