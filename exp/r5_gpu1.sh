#!/bin/bash
# round 5, call 1: the round's new tests (engine vs pyref fixtures, threading contract, unit-records kernel, pairing chains without
# the earlier launches' expansions, traffic counters by launch index) and a default bench line with the KiB-correct traffic
cd "$(dirname "$0")/.."
O=gpurun_out/r5_1; mkdir -p $O
timeout 1500 python -m pytest tests/test_pyref_gpu.py tests/test_threads_gpu.py "tests/test_parity_gpu.py::test_unit_records_kernel_matches_reference_indexing" \
  "tests/test_parity_gpu.py::test_pairing_value_chain_does_not_depend_on_expansion" tests/test_bench_gpu.py -m gpu -q > $O/pytest_new.log 2>&1; echo "pytest new rc $?"; tail -15 $O/pytest_new.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; cat $O/bench.time | tail -3; tail -c 1500 $O/bench.json; tail -5 $O/bench.err
# where the digit chain's cycles go, per wave and per record count (stamped builds, built on the box)
( bash exp/wave_stamps.sh && python exp/wave_stamps.py 64 bn256 && python exp/wave_stamps.py 8 bn256 ) > $O/stamps_bn256.log 2>&1; tail -45 $O/stamps_bn256.log
( FPK=1 bash exp/wave_stamps.sh && python exp/wave_stamps.py 16 bls12_381 ) > $O/stamps_bls.log 2>&1; tail -40 $O/stamps_bls.log
