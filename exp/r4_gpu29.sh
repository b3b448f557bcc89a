#!/bin/bash
# round 4, call 29: the gate in front of an expansion that becomes ready together with the next stage's digit chain (single batches
# of 64 bn256 checks were 4.2 or 5.5 ms depending on which of the two the dispatcher saw first); H2E_SCHED=36 = without the gate
cd "$(dirname "$0")/.."
O=gpurun_out/r4_29; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', d['single_batch_ms'] and round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3) if 'value_chain_ms' in r else None, 'x', round(sum(r['expansion_ms']),3) if 'expansion_ms' in r else None, 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
N=H2E_SCHED=36
for rep in 1 2 3; do
bench bn64_r1_nogate_$rep $N -- --workload pairing_bn256 --ring 1 --latency-steps 0
bench bn64_r1_gate_$rep X=1 -- --workload pairing_bn256 --ring 1 --latency-steps 0
bench bn64_nogate_$rep $N -- --workload pairing_bn256
bench bn64_gate_$rep X=1 -- --workload pairing_bn256
done
for rep in 1 2; do
bench bls16_nogate_$rep $N -- --workload pairing_bls12_381
bench bls16_gate_$rep X=1 -- --workload pairing_bls12_381
bench bn8_nogate_$rep $N -- --workload pairing_bn256 --units 8
bench bn8_gate_$rep X=1 -- --workload pairing_bn256 --units 8
done
