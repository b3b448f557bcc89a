#!/bin/bash
# round 4, call 46: 64 x bn256 pipelined with the expansion's result cache off (H2E_TUNE=0,2: 4 KB instead of 19 KB of LDS per expansion
# workgroup - beside a chain workgroup's ~100 KB a CU then takes four of them, not three), alternating with the default
cd "$(dirname "$0")/.."
O=gpurun_out/r4_46; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3), 'x', round(sum(r['expansion_ms']),3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for rep in 1 2 3; do
bench bn64_xc_$rep X=1 -- --workload pairing_bn256
bench bn64_noxc_$rep H2E_TUNE=0,2,0,0,0,0 -- --workload pairing_bn256
done
bench bls64_xc X=1 -- --workload pairing_bls12_381 --units 64
bench bls64_noxc H2E_TUNE=0,2,0,0,0,0 -- --workload pairing_bls12_381 --units 64
