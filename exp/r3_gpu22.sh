#!/bin/bash
# where does a round's time go: timing-only variants of the field chain (wrong arithmetic, --no-check)
cd "$(dirname "$0")/.."
O=gpurun_out/r3w; mkdir -p $O
for v in ${VARIANTS:-"" mul4 lin6 both noops}; do
L=""; [ -n "$v" ] && L="exp/_dbg/libh2e_$v.so"
H2E_LIB=$L timeout 600 python bench.py --sub --suite main --workload pairing_bn256 --traffic off --no-cpu-baseline --ring 1 --latency-steps 0 --no-check > $O/bn_$v.json 2> $O/bn_$v.err
python -c "
import json; d=json.loads(open('$O/bn_$v.json').read().strip().splitlines()[-1]); print('variant [$v]', round(d['ms_per_step'],2), d['roofline']['value_chain_ms'], d['roofline']['expansion_ms'])" || tail -3 $O/bn_$v.err
done
