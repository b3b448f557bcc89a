#!/bin/bash
# round-3 profile set (run on the GPU box from the repo root): the default bench line (children, PMC traffic passes, CPU baseline)
# and a rocprofv3 kernel trace + stats of the three workloads' bench commands -> gpurun_out/r3_p/, published into profiles/ by
# exp/publish_profiles_r3.py
TAG=${1:-r3_p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
for wl in msm pairing_bn256 pairing_bls12_381; do
  rocprofv3 --kernel-trace --stats -d $OUT/stats_$wl -o run --output-format csv -- python3 bench.py --sub --suite main --workload $wl --no-cpu-baseline --traffic off > $OUT/stats_$wl.log 2>&1
done
# the value chain alone (nothing else in flight): one batch after the other
for wl in pairing_bn256 pairing_bls12_381; do
  rocprofv3 --kernel-trace --stats -d $OUT/stats_${wl}_ring1 -o run --output-format csv -- python3 bench.py --sub --suite main --workload $wl --ring 1 --latency-steps 0 --no-cpu-baseline --traffic off > $OUT/stats_${wl}_ring1.log 2>&1
done
rocprofv3 --kernel-trace --stats -d $OUT/stats_job -o run --output-format csv -- python3 bench.py --sub --suite main --workload msm --job-tiles 1024 --no-cpu-baseline --traffic off > $OUT/stats_job.log 2>&1
ls $OUT; cat $OUT/bench.time; tail -c 600 $OUT/bench.json
