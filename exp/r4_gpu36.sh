#!/bin/bash
# round 4, call 36: the whole -m gpu suite, then the profile set (exp/r4_profiles.sh r4_t) on the round's final code
cd "$(dirname "$0")/.."
O=gpurun_out/r4_36; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
bash exp/r4_profiles.sh r4_t > $O/profiles.log 2>&1; tail -5 $O/profiles.log
