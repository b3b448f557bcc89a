#!/bin/bash
# round 4, call 22: the MSM step with the expansion's LDS result cache (H2E_TUNE=0,3) now that the windows' and the loop's replays - the
# LDS-hungry neighbours it used to keep off the CUs - are hint stores; alternating with the default in one box, then its PMC traffic
cd "$(dirname "$0")/.."
O=gpurun_out/r4_22; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3), 'traffic', x.get('traffic'), 'alg', x.get('algorithmic_bytes_per_launch'))" || tail -3 $O/$tag.err
}
for rep in 1 2 3; do
bench msm_base_$rep X=1 -- --traffic off
bench msm_xc_$rep H2E_TUNE=0,3,0,0,0,0 -- --traffic off
done
bench msm_xc_traffic H2E_TUNE=0,3,0,0,0,0 -- --traffic auto --steps 10
bench job_xc H2E_TUNE=0,3,0,0,0,0 -- --traffic off --job-tiles 1024
bench job_base X=1 -- --traffic off --job-tiles 1024
