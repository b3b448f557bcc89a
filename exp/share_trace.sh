#!/bin/bash
# kernel traces of the 8-GPU share of the bn256 pairing batch (8 checks) at ring 4 and ring 8: bash exp/share_trace.sh -> gpurun_out/share_trace/
cd "$(dirname "$0")/.."
O=gpurun_out/share_trace; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for ring in 4 8; do
  timeout 400 rocprofv3 --kernel-trace -d $O/ring$ring -o run --output-format csv -- python3 bench.py --sub --suite main --traffic off --no-cpu-baseline --workload pairing_bn256 --units 8 --ring $ring --steps 24 --warmup 4 --latency-steps 0 > $O/ring$ring.log 2>&1
  grep -o '"ms_per_step": [0-9.]*' $O/ring$ring.log | head -1
done
for ring in 4 8; do
  timeout 300 python bench.py --sub --suite main --traffic off --no-cpu-baseline --workload pairing_bn256 --units 8 --ring $ring --latency-steps 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1
done
