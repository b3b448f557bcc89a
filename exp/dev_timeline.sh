#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py tests/test_threads_gpu.py -m gpu -x -q -k "small_batches or pairing_check or batch_16 or bench_small_pairing or digest or pipelined or threads or strong_shares" 2>&1 | tail -3
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
for ring in 16 24 32; do
for cfg in "pairing_bls12_381 2" "pairing_bn256 8" "pairing_bls12_381 16"; do
  set -- $cfg
  echo "ring $ring $1 x $2: $(timeout 300 $B --workload $1 --units $2 --ring $ring 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ')"
done
done
