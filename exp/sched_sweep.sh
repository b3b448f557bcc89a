#!/bin/bash
# H2E_SCHED sweep (which stream a pipelined run's small expansions / fix-ups use) x ring depth; run on the GPU box from the repo root
show='import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print(round(d["ms_per_step"],2), "chain", [round(x,1) for x in r["value_chain_ms"] if x>0.3], "x", [round(x,1) for x in r["expansion_ms"] if x>0.5])'
for sched in 0 1 3 0 1 3; do
  for ring in 2 3; do
    echo -n "sched $sched ring $ring: "
    H2E_SCHED=$sched python bench.py --steps 12 --warmup 4 --no-cpu-baseline --traffic off --ring $ring 2>/dev/null | python -c "$show"
  done
done
