#!/bin/bash
# round 4, call 6: a pairing check cut into several launches (H2E_PAIRING_SPLITS, record time): parity of every pairing test at each
# level, then single-batch latency and pipelined step of the three pairing batch sizes
cd "$(dirname "$0")/.."
O=gpurun_out/r4_6; mkdir -p $O
for sp in 0 1 2; do
H2E_PAIRING_SPLITS=$sp timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py tests/test_check_gpu.py -m gpu -x -q -k "pairing and not soak and not variants and not full_size" > $O/pytest_sp$sp.log 2>&1; echo "splits $sp pytest rc $?"; tail -3 $O/pytest_sp$sp.log
done
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', round(d['single_batch_ms'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.1], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.1], 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for sp in 0 1 2 3; do
bench bn64_r1_sp$sp H2E_PAIRING_SPLITS=$sp -- --workload pairing_bn256 --ring 1
bench bn64_r3_sp$sp H2E_PAIRING_SPLITS=$sp -- --workload pairing_bn256 --ring 3
bench bls16_r1_sp$sp H2E_PAIRING_SPLITS=$sp -- --workload pairing_bls12_381 --ring 1
bench bls16_r3_sp$sp H2E_PAIRING_SPLITS=$sp -- --workload pairing_bls12_381 --ring 3
bench bn8_r1_sp$sp H2E_PAIRING_SPLITS=$sp -- --workload pairing_bn256 --units 8 --ring 1
bench bls2_r1_sp$sp H2E_PAIRING_SPLITS=$sp -- --workload pairing_bls12_381 --units 2 --ring 1
done
