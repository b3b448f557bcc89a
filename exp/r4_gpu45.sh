#!/bin/bash
# round 4, call 45: the new small-batch parity test (packed expansion, ragged group sizes)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_45; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "small_batches_packed" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log
