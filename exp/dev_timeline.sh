#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r5_t
( time python bench.py > gpurun_out/r5_t/bench.json 2> gpurun_out/r5_t/bench.err ) 2> gpurun_out/r5_t/bench.time
tail -c 900 gpurun_out/r5_t/bench.json; cat gpurun_out/r5_t/bench.time
