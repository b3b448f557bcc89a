#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3s; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "pairing or digest or ops or tower" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for w in pairing_bn256 pairing_bls12_381; do
for r in 1 3; do
timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --ring $r --latency-steps 0 > $O/${w}_ring$r.json 2> $O/${w}_ring$r.err
python -c "
import json; d=json.loads(open('$O/${w}_ring$r.json').read().strip().splitlines()[-1]); print('$w', $r, d['ms_per_step'], d['roofline']['value_chain_ms'], d['roofline']['expansion_ms'])"
done; done
