#!/bin/bash
# round 4, call 39: the MSM run ends with four small segments whose expansions and one-workgroup inverse fix-ups alternate on one stream
# (4.3 ms, profiles/r4_t_msm: 12.97 -> 17.3) and the slot's next run starts when they are through.  H2E_SCHED=68: those fix-ups on the
# common fix-up stream instead
cd "$(dirname "$0")/.."
O=gpurun_out/r4_39; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
S=H2E_SCHED=68
bench warm X=1 -- --workload msm
for rep in 1 2 3; do
bench msm_s4_$rep X=1 -- --workload msm
bench msm_s68_$rep $S -- --workload msm
done
bench job_s4 X=1 -- --workload msm --job-tiles 1024
bench job_s68 $S -- --workload msm --job-tiles 1024
bench bls16_s4 X=1 -- --workload pairing_bls12_381
bench bls16_s68 $S -- --workload pairing_bls12_381
bench bn8_s4 X=1 -- --workload pairing_bn256 --units 8
bench bn8_s68 $S -- --workload pairing_bn256 --units 8
