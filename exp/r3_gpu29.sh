#!/bin/bash
# A/B in one box, alternating: the previous commit (15 computing waves, rows store their hints themselves) vs this one (14 + storing wave)
cd "$(dirname "$0")/.."
O=gpurun_out/r3_ab; mkdir -p $O
for i in 1 2 3; do
for v in prefetch ""; do
L=""; [ -n "$v" ] && L="exp/_dbg/libh2e_$v.so"
for w in pairing_bn256 pairing_bls12_381; do
H2E_LIB=$L timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --latency-steps 2 > $O/x.json 2> $O/x.err
python -c "
import json; d=json.loads(open('$O/x.json').read().strip().splitlines()[-1]); print('[$v] $w', round(d['ms_per_step'],2), round(d['single_batch_ms'],2), [round(x,2) for x in d['roofline']['value_chain_ms']], [round(x,2) for x in d['roofline']['expansion_ms']])" || tail -3 $O/x.err
done; done; done
