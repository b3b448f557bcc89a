#!/bin/bash
# pairing steady state: expansion waves at normal priority (chain waves keep s_setprio 3)?
cd "$(dirname "$0")/.."
O=gpurun_out/r3z; mkdir -p $O
for t in "0,2,0,0,0" "0,0,0,0,0"; do
for w in pairing_bn256 pairing_bls12_381; do
H2E_TUNE="$t" timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --latency-steps 0 > $O/x.json 2> $O/x.err
python -c "
import json; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('tune $t', '$w', round(d['ms_per_step'],3), d['roofline']['value_chain_ms'], d['roofline']['expansion_ms'])"
done; done
