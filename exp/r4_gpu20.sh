#!/bin/bash
# round 4, call 20: op-cache eviction test; the 64-check pairing batches with the expansion's LDS result cache (H2E_TUNE=0,3) and with
# other sub-range lengths (H2E_PAIRING_CUT), alternating in one box
cd "$(dirname "$0")/.."
O=gpurun_out/r4_20; mkdir -p $O
timeout 1500 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0 "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.3], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.1], 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for rep in 1 2 3; do
bench bn64_base_$rep X=1 -- --workload pairing_bn256
bench bn64_xc_$rep H2E_TUNE=0,3,0,0,0,0 -- --workload pairing_bn256
bench bn64_cut12_$rep H2E_PAIRING_CUT=12 -- --workload pairing_bn256
bench bn64_cut8xc_$rep H2E_PAIRING_CUT=8 H2E_TUNE=0,3,0,0,0,0 -- --workload pairing_bn256
done
for rep in 1 2; do
bench bls16_base_$rep X=1 -- --workload pairing_bls12_381
bench bls16_cut16_$rep H2E_PAIRING_CUT=16 -- --workload pairing_bls12_381
bench bls16_cut6_$rep H2E_PAIRING_CUT=6 -- --workload pairing_bls12_381
done
