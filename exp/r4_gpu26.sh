#!/bin/bash
# round 4, call 26: with the packed expansion shorter, how many runs in flight and how many stages per pairing check pay now?
# ring 3 / 4 / 6 and H2E_PAIRING_SPLITS 1 / 2 / 3 for 16 x bls12_381; splits for 64 x bn256; BASELINE's 8-GPU shares pipelined
cd "$(dirname "$0")/.."
O=gpurun_out/r4_26; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for rep in 1 2; do
bench bls16_ring3_$rep X=1 -- --workload pairing_bls12_381 --ring 3
bench bls16_ring4_$rep X=1 -- --workload pairing_bls12_381 --ring 4
bench bls16_ring6_$rep X=1 -- --workload pairing_bls12_381 --ring 6
bench bls16_splits2_$rep H2E_PAIRING_SPLITS=2 -- --workload pairing_bls12_381
bench bls16_splits3_$rep H2E_PAIRING_SPLITS=3 -- --workload pairing_bls12_381
bench bn64_splits1_$rep X=1 -- --workload pairing_bn256
bench bn64_splits2_$rep H2E_PAIRING_SPLITS=2 -- --workload pairing_bn256
bench bn64_splits3_$rep H2E_PAIRING_SPLITS=3 -- --workload pairing_bn256
done
bench bn8_ring6 X=1 -- --workload pairing_bn256 --units 8 --ring 6
bench bls2_ring3 X=1 -- --workload pairing_bls12_381 --units 2 --ring 3
bench bls2_ring6 X=1 -- --workload pairing_bls12_381 --units 2 --ring 6
bench bls2_ring12 X=1 -- --workload pairing_bls12_381 --units 2 --ring 12
