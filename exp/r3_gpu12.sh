#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3l; mkdir -p $O
python exp/wave_stamps.py > $O/stamps.txt 2>&1; cat $O/stamps.txt | tail -9
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "pairing or digest or ops or tower" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
H2E_FIELD_CHAIN=lanes timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing" > $O/pytest_lanes.log 2>&1; echo "pytest lanes rc $?"; tail -3 $O/pytest_lanes.log
for w in pairing_bn256 pairing_bls12_381; do
for r in 1 2 3; do
timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --ring $r --latency-steps 0 > $O/${w}_ring$r.json 2> $O/${w}_ring$r.err
python -c "
import json; d=json.loads(open('$O/${w}_ring$r.json').read().strip().splitlines()[-1]); print('$w', $r, d['ms_per_step'], d['roofline']['value_chain_ms'], d['roofline']['expansion_ms'])"
done; done
