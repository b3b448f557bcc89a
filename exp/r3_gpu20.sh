#!/bin/bash
# CUs kept free of the expansion streams (the chain streams may use all): does the starved front of the next run's chain get through?
cd "$(dirname "$0")/.."
O=gpurun_out/r3u; mkdir -p $O
for cfg in "0,0,0,0" "8,0,1,1" "16,0,1,1" "32,0,1,1" "16,2,1,1" "16,1,1,1"; do
H2E_CU_RESERVE="$cfg" timeout 600 python bench.py --suite main --traffic off --no-cpu-baseline --latency-steps 0 > $O/msm.json 2> $O/msm.err
python -c "
import json; d=json.loads(open('$O/msm.json').read().strip().splitlines()[-1]); r=d['roofline']; print('cu $cfg', round(d['ms_per_step'],2), 'x', [round(x,1) for x in r['expansion_ms'] if x>0.5], 'chain', [round(x,1) for x in r['value_chain_ms'] if x>0.5])" || tail -3 $O/msm.err
done
