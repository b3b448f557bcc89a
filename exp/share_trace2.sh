#!/bin/bash
# the 8-check share with NO host waits in the loop (exp/submit_host_time.py, profiling off): how many chains does the GPU run at once?
cd "$(dirname "$0")/.."
O=gpurun_out/share_trace2; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for ring in ${RINGS:-8}; do
  rm -rf $O/ring$ring
  timeout 400 rocprofv3 --kernel-trace -d $O/ring$ring -o run --output-format csv -- python3 exp/submit_host_time.py ${CURVE:-bn256} ${UNITS:-8} $ring 0 > $O/ring$ring.log 2>&1
  tail -1 $O/ring$ring.log
done
