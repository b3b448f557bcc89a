#!/bin/bash
# round-6 profile set (run on the GPU box from the repo root): the default bench line as the driver runs it (headline + bench_detail.json:
# children, PMC traffic passes, CPU baseline), a rocprofv3 kernel trace + stats of every workload's bench command, the PMC traffic of the
# packed expansions (8 / 2 / 16 checks), the consumer-ready runs (two passes / first pass) -> gpurun_out/<tag>/, published into profiles/
# by exp/publish_profiles_r6.py.  The set records the hash of the sources it was taken from (exp/source_hash.py): publish refuses a set
# whose hash is not the tree's.
TAG=${1:-r6_p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python exp/source_hash.py > $OUT/source_hash.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail-file $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time
prof() {  # tag, bench args
t=$1; shift
rocprofv3 --kernel-trace --stats -d $OUT/stats_$t -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --full-line "$@" > $OUT/stats_$t.log 2>&1
}
prof msm --workload msm
prof pairing_bn256 --workload pairing_bn256
prof pairing_bls12_381 --workload pairing_bls12_381
# one batch after the other: the value chain and the expansion without other runs beside them
prof pairing_bn256_ring1 --workload pairing_bn256 --ring 1 --latency-steps 0
prof pairing_bls12_381_ring1 --workload pairing_bls12_381 --ring 1 --latency-steps 0
prof job --workload msm --job-tiles 1024
# the consumer-ready runs: run + export (two passes), and the columns straight out of the expansion (h2e_run_tape_cols)
prof consumer_ready --workload msm --ring 1 --steps 3 --warmup 1 --latency-steps 0 --consumer-ready 3
# PMC traffic of the packed expansions (round 5's review: "not measured" for the shares): the bench's own two --pmc child passes
for cfg in "pairing_bn256 8" "pairing_bls12_381 2" "pairing_bls12_381 16"; do
  set -- $cfg
  timeout 900 python bench.py --sub --suite main --workload $1 --units $2 --steps 60 --warmup 5 --no-cpu-baseline --full-line > $OUT/traffic_$1_$2.json 2> $OUT/traffic_$1_$2.err
done
# run-to-run spread of the headline lines (three fresh processes each)
for w in msm pairing_bn256 pairing_bls12_381; do
for i in 1 2 3; do
timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --full-line > $OUT/spread_${w}_$i.json 2> $OUT/spread_${w}_$i.err
done
done
ls $OUT | head -60; cat $OUT/bench.time; tail -c 1500 $OUT/bench.json
