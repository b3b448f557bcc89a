#!/bin/bash
# persistent-form expansion: workgroups per CU sweep on the MSM headline
cd "$(dirname "$0")/.."
O=gpurun_out/r3r; mkdir -p $O
for pw in 0 8 6 5 4 3; do
H2E_TUNE="0,2,0,0,$pw" timeout 600 python bench.py --suite main --traffic off --no-cpu-baseline --latency-steps 2 > $O/msm_$pw.json 2> $O/msm_$pw.err
python -c "
import json; d=json.loads(open('$O/msm_$pw.json').read().strip().splitlines()[-1]); r=d['roofline']; print('pw $pw', round(d['ms_per_step'],2), 'single', round(d['single_batch_ms'],2), 'x', [round(x,1) for x in r['expansion_ms'] if x>0.5], 'chain', [round(x,1) for x in r['value_chain_ms'] if x>0.5])" || tail -3 $O/msm_$pw.err
done
