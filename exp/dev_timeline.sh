#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_threads_gpu.py tests/test_bench_gpu.py -m gpu -x -q -k "pipelined or batch_64 or threads or bench" 2>&1 | tail -2
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
for rep in 1 2; do
  echo "bn256 x 64: $(timeout 300 $B --workload pairing_bn256 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  echo "msm: $(timeout 300 $B --workload msm 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  echo "bn256 x 32: $(timeout 300 $B --workload pairing_bn256 --units 32 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  echo "bn256 x 32 ring 4: $(timeout 300 $B --workload pairing_bn256 --units 32 --ring 4 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
  echo "bn256 x 64 ring 6: $(timeout 300 $B --workload pairing_bn256 --ring 6 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')"
done
