#!/bin/bash
# round 4, call 43: final code (small fix-ups of runs without a big expansion on the fix-up stream; pairing ring 4): the whole -m gpu
# suite and the profile set (r4_u)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_43; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
bash exp/r4_profiles.sh r4_u > $O/profiles.log 2>&1; tail -3 $O/profiles.log | cut -c1-1500
