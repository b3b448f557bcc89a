#!/bin/bash
# round 4, call 40: small fix-ups on the fix-up stream (H2E_SCHED=68) for the small pairing batches, two more rounds + ring 4
cd "$(dirname "$0")/.."
O=gpurun_out/r4_40; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
S=H2E_SCHED=68
for rep in 1 2; do
bench bls16_s4_$rep X=1 -- --workload pairing_bls12_381
bench bls16_s68_$rep $S -- --workload pairing_bls12_381
bench bls16_s68_ring4_$rep $S -- --workload pairing_bls12_381 --ring 4
bench bn8_s4_$rep X=1 -- --workload pairing_bn256 --units 8
bench bn8_s68_$rep $S -- --workload pairing_bn256 --units 8
bench bls2_s4_$rep X=1 -- --workload pairing_bls12_381 --units 2
bench bls2_s68_$rep $S -- --workload pairing_bls12_381 --units 2
bench bn64_s4_$rep X=1 -- --workload pairing_bn256
bench bn64_s68_$rep $S -- --workload pairing_bn256
done
