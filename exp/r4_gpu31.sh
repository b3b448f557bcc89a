#!/bin/bash
# round 4, call 31: the pairings' pipelined step is the expansion stream's (two expansions per run back to back, each 15-40 % slower
# next to the chains of the next runs).  Does it pay to leave the expansion's waves at normal priority (H2E_TUNE second field 1
# instead of 3) so that the chain's rounds get their issue slots first - and then a deeper ring?
cd "$(dirname "$0")/.."
O=gpurun_out/r4_31; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
P=H2E_TUNE=0,1,0,0,0,0
for rep in 1 2; do
bench bn64_prio3_$rep X=1 -- --workload pairing_bn256
bench bn64_prio0_$rep $P -- --workload pairing_bn256
bench bn64_prio0_ring4_$rep $P -- --workload pairing_bn256 --ring 4
bench bn64_prio3_ring4_$rep X=1 -- --workload pairing_bn256 --ring 4
bench bls16_prio3_$rep X=1 -- --workload pairing_bls12_381
bench bls16_prio0_$rep $P -- --workload pairing_bls12_381
done
