#!/bin/bash
# the whole -m gpu suite, the round's profile set and the device timelines on the tree as it is: bash exp/final_check.sh <profile tag>
cd "$(dirname "$0")/.."
TAG=${1:-r5_t}
O=gpurun_out/final_$TAG; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -rs > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
bash exp/r5_profiles.sh $TAG > $O/profiles.log 2>&1; tail -3 $O/profiles.log | cut -c1-1200
# device timelines (debug build's stamp kernels: no profiler in the way of the host) of the pipelined lines
if [ -f exp/_dbg/libh2e_dbg.so ]; then
  W="bash exp/with_lib.sh exp/_dbg/libh2e_dbg.so --"   # (the debug build stands in for the product only while one command runs)
  B="--sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
  $W python exp/bench_timeline.py 30 18 -- $B --workload pairing_bn256 --units 8 2>&1 | grep -v "^{" > $O/dev_timeline_bn256_share8.txt
  $W python exp/bench_timeline.py 30 18 -- $B --workload pairing_bls12_381 --units 2 2>&1 | grep -v "^{" > $O/dev_timeline_bls12_381_share8.txt
  $W python exp/bench_timeline.py 30 18 -- $B --workload pairing_bls12_381 2>&1 | grep -v "^{" > $O/dev_timeline_bls12_381.txt
  $W python exp/bench_timeline.py 24 8 -- $B --workload pairing_bn256 2>&1 | grep -v "^{" > $O/dev_timeline_bn256.txt
  $W python exp/bench_timeline.py 20 6 -- $B --workload msm --steps 16 2>&1 | grep -v "^{" > $O/dev_timeline_msm.txt
  wc -l $O/dev_timeline_*.txt
fi
