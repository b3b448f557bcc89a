#!/bin/bash
# HBM traffic of the column-emission kernel (h2e_run_tape_cols) and of the export beside it from the PMC counters: two rocprofv3 --pmc passes
# (WRITE_SIZE, FETCH_SIZE: KiB; FETCH_SIZE x 2 on gfx950 - MI355X_MICROARCH.md) of bench.py's consumer-ready run, one repetition.
# -> gpurun_out/<tag>/cols_pmc.txt (exp/r6_cols_pmc_summary.py)
TAG=${1:-r6_cols_pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
python exp/source_hash.py > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  timeout 600 rocprofv3 --pmc $c -d $OUT/pmc_$c -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --sub --suite main --workload msm --ring 1 \
    --steps 1 --warmup 0 --latency-steps 0 --consumer-ready 1 --traffic off --no-cpu-baseline --full-line > $OUT/pmc_$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python exp/r6_cols_pmc_summary.py $OUT > $OUT/cols_pmc.txt 2>&1
cat $OUT/cols_pmc.txt
# the raw counter files are big: keep the summary only
rm -rf $OUT/pmc_WRITE_SIZE $OUT/pmc_FETCH_SIZE
