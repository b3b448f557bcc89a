#!/bin/bash
# labelled kernel timeline of a pipelined bench run: the debug build of the C-ABI layer logs every engine call (run, segment, stream),
# exp/trace_labelled.py matches the log with a rocprofv3 kernel trace in dispatch order -> gpurun_out/<tag>/labelled.txt
# usage (GPU box, repo root): bash exp/trace_labelled.sh <tag> [bench args]
TAG=${1:-tl}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
bash exp/build_dbg.sh > $OUT/build.log 2>&1 || { tail -5 $OUT/build.log; exit 1; }
rm -f /tmp/h2e_dbg.log
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
export H2E_DEBUG_LOG=/tmp/h2e_dbg.log
bash exp/with_lib.sh exp/_dbg/libh2e_dbg.so -- rocprofv3 --kernel-trace -d $OUT/trace -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --latency-steps 0 --steps ${STEPS:-8} --warmup ${WARM:-2} "$@" > $OUT/bench.log 2>&1
unset H2E_DEBUG_LOG
cp /tmp/h2e_dbg.log $OUT/launch_log.txt
python exp/trace_labelled.py $OUT/trace/run_kernel_trace.csv $OUT/launch_log.txt ${RUNS:-3} > $OUT/labelled.txt 2> $OUT/labelled.err
tail -2 $OUT/labelled.err; grep -o '"ms_per_step": [0-9.]*' $OUT/bench.log | head -1; wc -l $OUT/labelled.txt
