#!/bin/bash
# round 4, call 14: the digit-row chain's hint log (a round's hint values leave the CU as one contiguous run): every pairing parity test,
# the chain next to a fill / copy kernel again, and the pairing lines
cd "$(dirname "$0")/.."
O=gpurun_out/r4_14; mkdir -p $O
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py tests/test_check_gpu.py tests/test_digit_rows_gpu.py -m gpu -x -q -k "pairing or tower or digit or integer_chip" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 600 python exp/chain_vs_fill.py 64 > $O/chain_vs_fill_64.txt 2>&1; tail -4 $O/chain_vs_fill_64.txt
timeout 600 python exp/chain_vs_fill.py 8 > $O/chain_vs_fill_8.txt 2>&1; tail -4 $O/chain_vs_fill_8.txt
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.3], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.3], 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for rep in 1 2; do
bench bn64_$rep X=1 -- --workload pairing_bn256
bench bls16_$rep X=1 -- --workload pairing_bls12_381
bench bn64_r1_$rep X=1 -- --workload pairing_bn256 --ring 1
bench bls16_r1_$rep X=1 -- --workload pairing_bls12_381 --ring 1
done
bench bn8 X=1 -- --workload pairing_bn256 --units 8
bench bls2 X=1 -- --workload pairing_bls12_381 --units 2
bench bn64_sp2 H2E_PAIRING_SPLITS=2 -- --workload pairing_bn256
bench bn64_sp0 H2E_PAIRING_SPLITS=0 -- --workload pairing_bn256
