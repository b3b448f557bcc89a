#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3y; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "soak" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log | cut -c1-300
