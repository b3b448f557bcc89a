#!/bin/bash
# kernel timeline of the last bench step under one env setting: exp/trace_env.sh TAG VAR=value
TAG=$1; export "$2"
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $OUT/tr -o run --output-format csv -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $OUT/log 2>&1
python exp/timeline.py $(find $OUT/tr -name '*kernel_trace.csv' | head -1) 6 > $OUT/timeline.txt
tail -1 $OUT/log | cut -c1-200
