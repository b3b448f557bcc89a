#!/bin/bash
# round 4, call 52: the packed expansion capped at 256 registers (two waves per SIMD; exp/ab/libh2e_xp2.so = -DH2E_XP_WAVES=2) again, now
# that its waves are the order tables' (a wave alone on its SIMD issues one VALU instruction per 30 cycles: profiles/r4_packed_pmc.txt)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_52; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x', round(sum(r['expansion_ms']),3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
W=H2E_LIB=$PWD/exp/ab/libh2e_xp2.so
for rep in 1 2; do
bench bls16_r1_base_$rep X=1 -- --workload pairing_bls12_381 --ring 1 --latency-steps 0
bench bls16_r1_xp2_$rep $W -- --workload pairing_bls12_381 --ring 1 --latency-steps 0
bench bls16_base_$rep X=1 -- --workload pairing_bls12_381
bench bls16_xp2_$rep $W -- --workload pairing_bls12_381
bench bn32_r1_base_$rep X=1 -- --workload pairing_bn256 --units 32 --ring 1 --latency-steps 0
bench bn32_r1_xp2_$rep $W -- --workload pairing_bn256 --units 32 --ring 1 --latency-steps 0
done
