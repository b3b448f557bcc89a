#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3x; mkdir -p $O
timeout 1500 python -m pytest tests/test_digit_rows_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -30 $O/pytest.log | cut -c1-300
bash exp/r3_gpu18.sh
