#!/bin/bash
# round-5 profile set (run on the GPU box from the repo root): the default bench line (children, PMC traffic passes, CPU baseline) and a
# rocprofv3 kernel trace + stats of every workload's bench command -> gpurun_out/<tag>/, published into profiles/ by
# exp/publish_profiles_r5.py
TAG=${1:-r5_p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
prof() {  # tag, bench args
t=$1; shift
rocprofv3 --kernel-trace --stats -d $OUT/stats_$t -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off "$@" > $OUT/stats_$t.log 2>&1
}
prof msm --workload msm
prof pairing_bn256 --workload pairing_bn256
prof pairing_bls12_381 --workload pairing_bls12_381
# one batch after the other: the value chain and the expansion without other runs beside them
prof pairing_bn256_ring1 --workload pairing_bn256 --ring 1 --latency-steps 0
prof pairing_bls12_381_ring1 --workload pairing_bls12_381 --ring 1 --latency-steps 0
# one GPU's share of configs[3] / configs[4] at 8 GPUs
prof pairing_bn256_share8_ring1 --workload pairing_bn256 --units 8 --ring 1 --latency-steps 0
prof pairing_bls12_381_share8_ring1 --workload pairing_bls12_381 --units 2 --ring 1 --latency-steps 0
prof job --workload msm --job-tiles 1024
# run-to-run spread of the headline lines (three fresh processes each)
for w in msm pairing_bn256 pairing_bls12_381; do
for i in 1 2 3; do
timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline > $OUT/spread_${w}_$i.json 2> $OUT/spread_${w}_$i.err
done
done
ls $OUT | head -50; cat $OUT/bench.time; tail -c 900 $OUT/bench.json
