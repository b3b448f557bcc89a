#!/bin/bash
# round 5, call 15: the whole -m gpu suite and the profile set (r5_q) on the round's final code
cd "$(dirname "$0")/.."
O=gpurun_out/r5_15; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
bash exp/r5_profiles.sh r5_q > $O/profiles.log 2>&1; tail -3 $O/profiles.log | cut -c1-1200
