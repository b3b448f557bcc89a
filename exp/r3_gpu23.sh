#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3x; mkdir -p $O
timeout 1500 python -m pytest tests/test_digit_rows_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest.log
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing_check" > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -3 $O/pytest2.log
