#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py -m gpu -x -q -k "small_batches or pairing_check_bls or batch_16 or bench_small_pairing or digest" 2>&1 | tail -3
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2; do
for cfg in "pairing_bls12_381 16" "pairing_bls12_381 2" "pairing_bn256 8" "pairing_bn256 64"; do
  set -- $cfg
  echo "$1 x $2: $(timeout 300 $B --workload $1 --units $2 2>/dev/null | grep -o '"ms_per_step": [0-9.]*\|"single_batch_ms": [0-9.]*' | tr '\n' ' ')"
done
done
for cfg in "pairing_bls12_381 16" "pairing_bn256 8"; do
  set -- $cfg
  echo "ring 1 $1 x $2: $(timeout 300 $B --workload $1 --units $2 --ring 1 --latency-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['ms_per_step'],3), [round(x,3) for x in r['expansion_ms']], [round(x,3) for x in r['value_chain_ms']])")"
done
