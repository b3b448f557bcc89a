#!/bin/bash
cd "$(dirname "$0")/.."
bash exp/trace.sh r3o_trace --suite main
tail -120 gpurun_out/r3o_trace/timeline.txt
