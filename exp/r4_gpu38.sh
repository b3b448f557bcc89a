#!/bin/bash
# round 4, call 38: bench.py queues step k + 1 (behind a stream-side wait for step k - 1) BEFORE the host asks for step k - 1's launch
# times, with one job slot more than runs in flight - against the old order (BENCH_ORDER=block: host-side wait first), in one box
cd "$(dirname "$0")/.."
O=gpurun_out/r4_38; mkdir -p $O
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
B=BENCH_ORDER=block
for rep in 1 2 3; do
bench msm_block_$rep $B -- --workload msm
bench msm_late_$rep X=1 -- --workload msm
done
bench job_block $B -- --workload msm --job-tiles 1024
bench job_late X=1 -- --workload msm --job-tiles 1024
for w in "pairing_bn256 64" "pairing_bls12_381 16" "pairing_bn256 8" "pairing_bls12_381 2"; do
set -- $w
bench ${1}_$2_block $B -- --workload $1 --units $2
bench ${1}_$2_late X=1 -- --workload $1 --units $2
done
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
