#!/bin/bash
# round 4, call 9: where a pipelined run's held-back candidates expansion goes (H2E_SCHED: 4 default = behind the windows' predictors on
# the shared expansion stream, +8 = on the small-expansion stream, +16 = not held back), the consumer-ready step with the assigned-only
# export, and the run-to-run spread of the pairing single-batch latency
cd "$(dirname "$0")/.."
O=gpurun_out/r4_9; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.3], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.3], 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3), 'consumer', d.get('consumer_ready_ms_per_step'))" || tail -3 $O/$tag.err
}
for rep in 1 2; do
for sc in 4 12 20 28; do
bench msm_sched${sc}_$rep H2E_SCHED=$sc --
done
done
bench consumer X=1 -- --workload msm --ring 1 --steps 3 --warmup 1 --latency-steps 0 --consumer-ready 3
for rep in 1 2 3; do
bench bn64_$rep X=1 -- --workload pairing_bn256
bench bls16_$rep X=1 -- --workload pairing_bls12_381
done
