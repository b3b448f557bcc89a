#!/bin/bash
# round 4, call 10: export that does not read unassigned cells (tests + consumer-ready step); runs in flight (--ring) for the pairing
# batches now that a check is two launches, full batches and the 8-GPU shares
cd "$(dirname "$0")/.."
O=gpurun_out/r4_10; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_shape_gpu.py -m gpu -x -q -k "export or fixed or columns" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.3], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.3], 'whole', round(d['whole_step']['frac'],3), 'consumer', d.get('consumer_ready_ms_per_step'))" || tail -3 $O/$tag.err
}
bench consumer X=1 -- --workload msm --ring 1 --steps 3 --warmup 1 --latency-steps 0 --consumer-ready 3
for ring in 2 3 4 6; do
bench bn64_ring$ring X=1 -- --workload pairing_bn256 --ring $ring --latency-steps 0
bench bls16_ring$ring X=1 -- --workload pairing_bls12_381 --ring $ring --latency-steps 0
done
for ring in 3 6 10 16; do
bench bn8_ring$ring X=1 -- --workload pairing_bn256 --units 8 --ring $ring --latency-steps 0 --steps 60
bench bls2_ring$ring X=1 -- --workload pairing_bls12_381 --units 2 --ring $ring --latency-steps 0 --steps 60
done
