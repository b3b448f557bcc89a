#!/bin/bash
cd "$(dirname "$0")/.."
O=$PWD/gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -6 $O/pytest.log
for wl in pairing_bn256 pairing_bls12_381; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $O/prof_$wl -o run --output-format csv -- python3 /root/repo/bench.py --workload $wl --ring 1 --steps 6 --warmup 2 --traffic off --no-cpu-baseline --latency-steps 0 > $O/${wl}_ring1.json 2> $O/${wl}_ring1.err)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1); echo "== $wl"; head -12 "$f" | cut -c1-200
done
