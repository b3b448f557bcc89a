#!/bin/bash
# the whole -m gpu suite and the round's profile set on the tree as it is: bash exp/final_check.sh <profile tag>
cd "$(dirname "$0")/.."
TAG=${1:-r5_q}
O=gpurun_out/final_$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
bash exp/r5_profiles.sh $TAG > $O/profiles.log 2>&1; tail -3 $O/profiles.log | cut -c1-1200
