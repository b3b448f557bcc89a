#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_s
( time python bench.py > gpurun_out/r5_s/bench.json 2> gpurun_out/r5_s/bench.err ) 2> gpurun_out/r5_s/bench.time
tail -c 900 gpurun_out/r5_s/bench.json; cat gpurun_out/r5_s/bench.time
