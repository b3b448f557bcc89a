#!/bin/bash
# round 4, call 32: in the pipelined small-batch runs every expansion AND its inverse fix-up (one workgroup, 0.12 ms) sit on the one shared
# small-expansion stream (profiles/r4_s: x, fix-up, x, fix-up ... = the step).  H2E_SCHED=5: the fix-ups on the slot's side stream
cd "$(dirname "$0")/.."
O=gpurun_out/r4_32; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
S=H2E_SCHED=5
for rep in 1 2; do
bench bls16_s4_$rep X=1 -- --workload pairing_bls12_381
bench bls16_s5_$rep $S -- --workload pairing_bls12_381
bench bn8_s4_$rep X=1 -- --workload pairing_bn256 --units 8
bench bn8_s5_$rep $S -- --workload pairing_bn256 --units 8
bench bls2_s4_$rep X=1 -- --workload pairing_bls12_381 --units 2
bench bls2_s5_$rep $S -- --workload pairing_bls12_381 --units 2
bench msm_s4_$rep X=1 -- --workload msm
bench msm_s5_$rep $S -- --workload msm
done
bench bn64_s4 X=1 -- --workload pairing_bn256
bench bn64_s5 $S -- --workload pairing_bn256
bench bls16_s5_ring4 $S -- --workload pairing_bls12_381 --ring 4
bench bls16_s5_ring6 $S -- --workload pairing_bls12_381 --ring 6
