#!/bin/bash
# round 4, call 28: the whole -m gpu suite, then the profile set (exp/r4_profiles.sh r4_s) on the round's final code
cd "$(dirname "$0")/.."
O=gpurun_out/r4_28; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
bash exp/r4_profiles.sh r4_s > $O/profiles.log 2>&1; tail -5 $O/profiles.log
