#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3q; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -4 $O/pytest.log
bash exp/r3_profiles.sh r3_p 2>&1 | tail -12
