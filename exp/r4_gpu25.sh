#!/bin/bash
# round 4, call 25: the packed expansion with order tables (sub-ranges of one opcode sequence share a wave, heaviest waves first) against
# the same kernel taking the sub-ranges in tape order (H2E_TUNE's sixth field = 2), alternating in one box; parity first
cd "$(dirname "$0")/.."
O=gpurun_out/r4_25; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py tests/test_check_gpu.py -m gpu -x -q -k "not full_size and not batch_64_tiles" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
T=H2E_TUNE=0,3,0,0,0,2
for rep in 1 2; do
bench bls16_r1_tape_$rep $T -- --workload pairing_bls12_381 --ring 1 --latency-steps 0
bench bls16_r1_order_$rep X=1 -- --workload pairing_bls12_381 --ring 1 --latency-steps 0
bench bls16_tape_$rep $T -- --workload pairing_bls12_381
bench bls16_order_$rep X=1 -- --workload pairing_bls12_381
bench bn8_r1_tape_$rep $T -- --workload pairing_bn256 --units 8 --ring 1 --latency-steps 0
bench bn8_r1_order_$rep X=1 -- --workload pairing_bn256 --units 8 --ring 1 --latency-steps 0
bench bls2_r1_tape_$rep $T -- --workload pairing_bls12_381 --units 2 --ring 1 --latency-steps 0
bench bls2_r1_order_$rep X=1 -- --workload pairing_bls12_381 --units 2 --ring 1 --latency-steps 0
done
bench bn8_tape $T -- --workload pairing_bn256 --units 8
bench bn8_order X=1 -- --workload pairing_bn256 --units 8
bench bn32_r1_tape $T -- --workload pairing_bn256 --units 32 --ring 1 --latency-steps 0
bench bn32_r1_order X=1 -- --workload pairing_bn256 --units 32 --ring 1 --latency-steps 0
