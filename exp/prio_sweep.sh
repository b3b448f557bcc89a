#!/bin/bash
# stream-priority / x-split sweep of the pipelined bench (run on the GPU box from the repo root)
for cfg in "0,0,0" "1,0,0" "1,-1,0" "1,-1,-1" "0,-1,0"; do
  for split in 45 0; do
    echo -n "prio $cfg split $split: "
    H2E_STREAM_PRIORITIES=$cfg H2E_X_SPLIT=$split python bench.py --steps 10 --warmup 4 --no-cpu-baseline --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],2), 'chain', [round(x,1) for x in r['value_chain_ms'] if x>0.3], 'x', [round(x,1) for x in r['expansion_ms'] if x>0.5])"
  done
done
