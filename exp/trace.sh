#!/bin/bash
# kernel trace of a short pipelined bench run -> gpurun_out/<tag>/timeline.txt (last two steps); run on the GPU box from the repo root
TAG=${1:-trace}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 bench.py --no-cpu-baseline --traffic off --steps ${STEPS:-6} --warmup ${WARM:-2} "$@" > $OUT/stats.log 2>&1
python exp/timeline.py $OUT/stats/run_kernel_trace.csv $(( (${STEPS:-6} + ${WARM:-2}) / 2 )) > $OUT/timeline.txt
grep '"metric"' $OUT/stats.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(round(d['ms_per_step'],2), 'chain', [round(x,1) for x in r['value_chain_ms'] if x>0.3], 'x', [round(x,1) for x in r['expansion_ms'] if x>0.5])"
