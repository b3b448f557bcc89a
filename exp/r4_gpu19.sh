#!/bin/bash
# round 4, call 19: hint log with the flush split (LDS read at the round's start, store behind its rows) against the build without the log
cd "$(dirname "$0")/.."
O=gpurun_out/r4_19; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "pairing or tower" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.3], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.1], 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
OLD=H2E_LIB=$PWD/exp/ab/libh2e_no_hint_log.so
for rep in 1 2; do
bench bn64_r1_new_$rep X=1 -- --workload pairing_bn256 --ring 1
bench bn64_r1_old_$rep $OLD -- --workload pairing_bn256 --ring 1
bench bn64_r3_new_$rep X=1 -- --workload pairing_bn256
bench bn64_r3_old_$rep $OLD -- --workload pairing_bn256
bench bls16_r1_new_$rep X=1 -- --workload pairing_bls12_381 --ring 1
bench bls16_r1_old_$rep $OLD -- --workload pairing_bls12_381 --ring 1
bench bls16_r3_new_$rep X=1 -- --workload pairing_bls12_381
bench bls16_r3_old_$rep $OLD -- --workload pairing_bls12_381
done
