#!/bin/bash
# round 4, call 42: is ring 5 worse than ring 4 because streams start sharing hardware queues?  GPU_MAX_HW_QUEUES=32, H2E_SCHED=68
cd "$(dirname "$0")/.."
O=gpurun_out/r4_42; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for ring in 4 5 6 8; do
bench bls16_q32_ring$ring H2E_SCHED=68 GPU_MAX_HW_QUEUES=32 -- --workload pairing_bls12_381 --ring $ring
bench bn8_q32_ring$ring H2E_SCHED=68 GPU_MAX_HW_QUEUES=32 -- --workload pairing_bn256 --units 8 --ring $ring
bench bls2_q32_ring$ring H2E_SCHED=68 GPU_MAX_HW_QUEUES=32 -- --workload pairing_bls12_381 --units 2 --ring $ring
done
bench bn64_q32_ring4 H2E_SCHED=68 GPU_MAX_HW_QUEUES=32 -- --workload pairing_bn256 --ring 4
bench msm_q32 GPU_MAX_HW_QUEUES=32 -- --workload msm
