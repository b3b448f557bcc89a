#!/bin/bash
# the window expansion's split point (percent of its sub-ranges in the first launch) once more, on this round's kernels
cd "$(dirname "$0")/.."
O=gpurun_out/r3_split; mkdir -p $O
for rep in 1 2; do
for pct in 45 30 60 20 70; do
H2E_X_SPLIT=$pct timeout 600 python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0 > $O/x.json 2> $O/x.err
python -c "
import json; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('split $pct', round(d['ms_per_step'],2))" || tail -3 $O/x.err
done; done
