#!/bin/bash
# round 4, call 41: small pairing batches, fix-ups on the fix-up stream (68) +/- the small expansions on the slots' own streams (70), rings 4 / 5
cd "$(dirname "$0")/.."
O=gpurun_out/r4_41; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for s in 68 70; do
for ring in 4 5; do
bench bls16_s${s}_ring$ring H2E_SCHED=$s -- --workload pairing_bls12_381 --ring $ring
bench bn8_s${s}_ring$ring H2E_SCHED=$s -- --workload pairing_bn256 --units 8 --ring $ring
bench bls2_s${s}_ring$ring H2E_SCHED=$s -- --workload pairing_bls12_381 --units 2 --ring $ring
done
done
bench bn64_s68_ring4 H2E_SCHED=68 -- --workload pairing_bn256 --ring 4
bench bn64_s4_ring4 H2E_SCHED=4 -- --workload pairing_bn256 --ring 4
