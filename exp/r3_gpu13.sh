#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3m; mkdir -p $O
for lds in 0 120000 158000; do
for w in pairing_bn256; do
for r in 2 3 4; do
H2E_FIELD_LDS=$lds timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --ring $r --latency-steps 0 > $O/${w}_ring$r.json 2> $O/${w}_ring$r.err
python -c "
import json; d=json.loads(open('$O/${w}_ring$r.json').read().strip().splitlines()[-1]); print('lds $lds', '$w', $r, d['ms_per_step'], d['roofline']['value_chain_ms'], d['roofline']['expansion_ms'])"
done; done; done
