#!/bin/bash
# round 4, call 5: (a) the hint store with batched term loads: parity + the MSM headline; (b) small pairing batches with the
# expansion's result cache in LDS (H2E_TUNE second field bit 0): operands from LDS instead of a load that waits for the wave's stores
cd "$(dirname "$0")/.."
O=gpurun_out/r4_5; mkdir -p $O
timeout 1800 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing_check or msm_tile or integer_chip" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', round(d['single_batch_ms'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.3], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.3], 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
bench msm_store_1 X=1 --
bench msm_store_2 X=1 --
for cut in 16 8; do
for w in "pairing_bn256 8 bn8" "pairing_bls12_381 16 bls16" "pairing_bls12_381 2 bls2" "pairing_bn256 64 bn64"; do
set -- $w
bench $3_cut${cut}_packed_xc H2E_PAIRING_CUT=$cut H2E_TUNE=0,3,0,0,0,0 -- --workload $1 --units $2 --ring 1
bench $3_cut${cut}_plain_xc H2E_PAIRING_CUT=$cut H2E_TUNE=0,3,0,0,0,1 -- --workload $1 --units $2 --ring 1
bench $3_cut${cut}_packed H2E_PAIRING_CUT=$cut -- --workload $1 --units $2 --ring 1
done
done
bench bn64_r3_xc H2E_TUNE=0,3,0,0,0,0 -- --workload pairing_bn256 --units 64 --ring 3
bench bn64_r3 X=1 -- --workload pairing_bn256 --units 64 --ring 3
bench bls16_r3_xc H2E_TUNE=0,3,0,0,0,0 -- --workload pairing_bls12_381 --units 16 --ring 3
bench msm_xc H2E_TUNE=0,3,0,0,0,0 --
