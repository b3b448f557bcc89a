#!/bin/bash
# round 4, call 30: CUs set aside for the pairings' digit chains (H2E_CU_RESERVE = n, fix-up side, pattern, chain on all CUs): the chain's
# workgroups then never wait for an expansion's grid to drain - at the price of the expansion's CUs
cd "$(dirname "$0")/.."
O=gpurun_out/r4_30; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for cfg in 0 64,0,0,0 64,0,1,0 64,0,0,1 64,0,1,1 128,0,1,1 32,0,1,1 0; do
bench bn64_cu_${cfg//,/_} H2E_CU_RESERVE=$cfg -- --workload pairing_bn256
done
for cfg in 0 16,0,1,1 32,0,1,1 64,0,1,1 0; do
bench bls16_cu_${cfg//,/_} H2E_CU_RESERVE=$cfg -- --workload pairing_bls12_381
done
