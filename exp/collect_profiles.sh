#!/bin/bash
# run on the GPU box from the repo root: the bench line (with the in-run PMC traffic passes and the CPU baseline) and a
# rocprofv3 kernel trace + stats of the same command, for profiles/<tag>_*
set -x
TAG=${1:-r2_a}
WORKLOAD=${2:-msm}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --workload $WORKLOAD > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 bench.py --workload $WORKLOAD --no-cpu-baseline --traffic off > $OUT/stats.log 2>&1
ls -la $OUT $OUT/stats | head -30
tail -c 1500 $OUT/bench.json
