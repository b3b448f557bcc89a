#!/bin/bash
# round 4, call 49: if the MSM step is half a run's latency with two 64-tile buffer sets, what do more, smaller sets give?  32 tiles x 4
# sets (220 GB) and 48 x 3 (247 GB), per-step and as the 2^20-point job (1024 tiles = 32 steps of 32)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_49; mkdir -p $O
bench() {  # tag -- args
tag=$1; shift; shift
timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'value', '%.4g' % d['value'], 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'whole', round(d['whole_step']['frac'],3), 'pts/s', d.get('msm_points_per_sec'))" || tail -3 $O/$tag.err
}
bench msm_64x2 -- --workload msm
bench msm_32x4 -- --workload msm --units 32 --ring 4
bench msm_32x3 -- --workload msm --units 32 --ring 3
bench msm_48x3 -- --workload msm --units 48 --ring 3
bench job_64x2 -- --workload msm --job-tiles 1024
bench job_32x4 -- --workload msm --job-tiles 1024 --units 32 --ring 4
