#!/bin/bash
# round 4, call 34: the pipelined small-batch step IS the one shared small-expansion stream (gate, x, fix-up, x, fix-up of every run
# behind each other, profiles: r4_33).  H2E_SCHED 2 / 3: small expansions (and fix-ups) on the slot's own side stream instead
cd "$(dirname "$0")/.."
O=gpurun_out/r4_34; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for s in 4 2 3; do
for ring in 3 4 6; do
bench bls16_s${s}_ring$ring H2E_SCHED=$s -- --workload pairing_bls12_381 --ring $ring
done
done
for s in 4 3; do
bench bn8_s${s}_ring3 H2E_SCHED=$s -- --workload pairing_bn256 --units 8 --ring 3
bench bn8_s${s}_ring6 H2E_SCHED=$s -- --workload pairing_bn256 --units 8 --ring 6
bench bls2_s${s}_ring3 H2E_SCHED=$s -- --workload pairing_bls12_381 --units 2 --ring 3
bench bls2_s${s}_ring6 H2E_SCHED=$s -- --workload pairing_bls12_381 --units 2 --ring 6
bench msm_s${s} H2E_SCHED=$s -- --workload msm
done
