#!/bin/bash
# round 4, call 48: the whole -m gpu suite on the final library + one default bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r4_48; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -c 900 $O/bench.json
