#!/bin/bash
# round 4, call 24: the expansion kernels that hold a SIMD alone because of their register count (packed bn256 260 + 4, packed bls12_381
# 304 + 48, plain bls12_381 299 + 43) capped at 256 registers - two waves per SIMD, 4 / 35 / 32 registers spilled to scratch
# (exp/ab/libh2e_w2.so = -DH2E_X_WAVES_WIDE=2 -DH2E_XP_WAVES=2), alternating with the default build in one box
cd "$(dirname "$0")/.."
O=gpurun_out/r4_24; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
W=H2E_LIB=$PWD/exp/ab/libh2e_w2.so
H2E_LIB=$PWD/exp/ab/libh2e_w2.so timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing" > $O/pytest_w2.log 2>&1; echo "pytest w2 rc $?"; tail -3 $O/pytest_w2.log
for rep in 1 2; do
bench bls16_r1_base_$rep X=1 -- --workload pairing_bls12_381 --ring 1 --latency-steps 0
bench bls16_r1_w2_$rep $W -- --workload pairing_bls12_381 --ring 1 --latency-steps 0
bench bls16_base_$rep X=1 -- --workload pairing_bls12_381
bench bls16_w2_$rep $W -- --workload pairing_bls12_381
bench bn8_r1_base_$rep X=1 -- --workload pairing_bn256 --units 8 --ring 1 --latency-steps 0
bench bn8_r1_w2_$rep $W -- --workload pairing_bn256 --units 8 --ring 1 --latency-steps 0
bench bls2_r1_base_$rep X=1 -- --workload pairing_bls12_381 --units 2 --ring 1 --latency-steps 0
bench bls2_r1_w2_$rep $W -- --workload pairing_bls12_381 --units 2 --ring 1 --latency-steps 0
bench bls64_r1_base_$rep X=1 -- --workload pairing_bls12_381 --units 64 --ring 1 --latency-steps 0
bench bls64_r1_w2_$rep $W -- --workload pairing_bls12_381 --units 64 --ring 1 --latency-steps 0
done
bench bn8_base X=1 -- --workload pairing_bn256 --units 8
bench bn8_w2 $W -- --workload pairing_bn256 --units 8
