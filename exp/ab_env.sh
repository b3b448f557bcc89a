#!/bin/bash
# A/B of one env knob within one box: exp/ab_env.sh VAR v1 v2 ...   (each value: one bench.py process, 10 steps)
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --steps 10 --warmup 4 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('$var=$v', 'ms/step %.2f' % j['ms_per_step'], 'X %.2f' % j['roofline']['launch_ms'], 'chain', ['%.1f' % x for x in j['roofline']['value_chain_ms'][-2:]])
"
done
