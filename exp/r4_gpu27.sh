#!/bin/bash
# round 4, call 27: the plain expansion's digest sums as dynamic LDS (19 KB per workgroup in a run without a digest: 8 waves per CU
# instead of the 6 that 25 KB allow), against the previous build, alternating in one box
cd "$(dirname "$0")/.."
O=gpurun_out/r4_27; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
P=H2E_LIB=$PWD/exp/ab/libh2e_prev.so
for rep in 1 2 3; do
bench msm_prev_$rep $P -- --workload msm
bench msm_new_$rep X=1 -- --workload msm
done
bench bn64_prev $P -- --workload pairing_bn256
bench bn64_new X=1 -- --workload pairing_bn256
bench job_prev $P -- --workload msm --job-tiles 1024
bench job_new X=1 -- --workload msm --job-tiles 1024
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "digest or msm_tile" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
