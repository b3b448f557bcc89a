#!/bin/bash
# the 8-GPU shares (8 bn256 / 2 bls12_381 checks) and the full pairing batches at several ring depths, + a quick parity subset
cd "$(dirname "$0")/.."
O=gpurun_out/${OUT:-share_lines}; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_threads_gpu.py -m gpu -x -q -k "${K:-pairing_check or msm_tile or threads or pipelined}" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for ring in ${RINGS:-4 6 8}; do
  python exp/submit_host_time.py bn256 8 $ring 0 2>&1 | tail -1
  python exp/submit_host_time.py bn256 8 $ring 1 2>&1 | tail -1
  python exp/submit_host_time.py bls12_381 2 $ring 1 2>&1 | tail -1
done
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
for ring in ${RINGS:-4 6 8}; do
  for cfg in "pairing_bn256 8" "pairing_bls12_381 2" "pairing_bn256 64" "pairing_bls12_381 16"; do
    set -- $cfg
    echo "$1 x $2 ring $ring: $(timeout 300 $B --workload $1 --units $2 --ring $ring 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
