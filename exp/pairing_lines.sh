#!/bin/bash
# the pairing parity / variant / digit-row / check tests, then the pairing bench lines (ring 1 = the chain alone, pipelined, the 8-GPU
# shares), twice each: OUT=<dir under gpurun_out> bash exp/pairing_lines.sh    (environment knobs of the caller apply to every line)
cd "$(dirname "$0")/.."
O=gpurun_out/${OUT:-pairing_lines}; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_pyref_gpu.py tests/test_digit_rows_gpu.py tests/test_check_gpu.py -m gpu -x -q -k "pairing or digit or check" > $O/pytest_pairing.log 2>&1; echo "pytest pairing rc $?"; tail -3 $O/pytest_pairing.log
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2; do
  for w in pairing_bn256 pairing_bls12_381; do
    timeout 300 $B --workload $w --ring 1 --latency-steps 0 > $O/${w}_ring1_$rep.json 2> $O/${w}_ring1_$rep.err
    timeout 300 $B --workload $w > $O/${w}_$rep.json 2> $O/${w}_$rep.err
  done
done
timeout 300 $B --workload pairing_bn256 --units 8 > $O/pairing_bn256_share8.json 2> $O/pairing_bn256_share8.err
timeout 300 $B --workload pairing_bls12_381 --units 2 > $O/pairing_bls12_381_share8.json 2> $O/pairing_bls12_381_share8.err
python exp/bench_lines.py $O
