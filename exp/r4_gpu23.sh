#!/bin/bash
# round 4, call 23: with the result cache on, an expansion workgroup needs 15-22 KB of LDS - so a chain kernel that asks for ~150 KB keeps
# its CU to itself (H2E_TUNE's first field; round 2 found it useless because the expansion then needed no LDS).  The pairings' digit chain
# and the MSM's small predictor grids, default vs reserved, alternating in one box
cd "$(dirname "$0")/.."
O=gpurun_out/r4_23; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
R=H2E_TUNE=150000,3,0,0,0,0
for rep in 1 2; do
bench bn64_base_$rep X=1 -- --workload pairing_bn256
bench bn64_res_$rep $R -- --workload pairing_bn256
bench bls16_base_$rep X=1 -- --workload pairing_bls12_381
bench bls16_res_$rep $R -- --workload pairing_bls12_381
bench msm_base_$rep X=1 -- --workload msm
bench msm_res_$rep $R -- --workload msm
done
bench bn64_r1_base X=1 -- --workload pairing_bn256 --ring 1
bench bn64_r1_res $R -- --workload pairing_bn256 --ring 1
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "msm" > $O/pytest_msm.log 2>&1; echo "pytest msm rc $?"; tail -3 $O/pytest_msm.log
