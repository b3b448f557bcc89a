#!/bin/bash
# round 4, call 44: with order tables a packed launch lasts as long as its heaviest sub-range - bn256 sub-ranges of 8 / 12 ops instead of 16
# (H2E_PAIRING_CUT) for the 8-check share and for the 64-check batch (whose hint stores grow with the cuts)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_44; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3), 'x', round(sum(r['expansion_ms']),3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for cut in 16 8 12; do
bench bn8_cut$cut H2E_PAIRING_CUT=$cut -- --workload pairing_bn256 --units 8
bench bn64_cut$cut H2E_PAIRING_CUT=$cut -- --workload pairing_bn256
done
bench bn8_cut16_b H2E_PAIRING_CUT=16 -- --workload pairing_bn256 --units 8
bench bn64_cut16_b H2E_PAIRING_CUT=16 -- --workload pairing_bn256
