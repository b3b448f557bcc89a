#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3_nostore; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing_check or soak" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for r in 1 3; do
for w in pairing_bn256 pairing_bls12_381; do
timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --ring $r --latency-steps 0 > $O/x.json 2> $O/x.err
python -c "
import json; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('$w ring $r', round(d['ms_per_step'],2), d['roofline']['value_chain_ms'], d['roofline']['expansion_ms'])" || tail -3 $O/x.err
done; done
