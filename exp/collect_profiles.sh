#!/bin/bash
# run on the GPU box from the repo root: bench line, rocprofv3 kernel stats / trace, HBM PMC passes of the headline workload
set -x
TAG=${1:-r1_d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_wr -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-check --no-cpu-baseline > $OUT/pmc_wr.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_rd -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-check --no-cpu-baseline > $OUT/pmc_rd.log 2>&1
ls -la $OUT $OUT/stats | head -30
tail -c 600 $OUT/bench.json
