#!/bin/bash
# round 5, call 7: the whole -m gpu suite on the round's code so far + per-kernel stats of the pairing batches run one after the other
# (where the value chain's bracket goes: digit chain | finalize | sinks | hint store)
cd "$(dirname "$0")/.."
O=gpurun_out/r5_7; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for w in pairing_bn256 pairing_bls12_381; do
  rocprofv3 --kernel-trace --stats -d $O/stats_$w -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --workload $w --ring 1 --latency-steps 0 --steps 10 --warmup 2 > $O/stats_$w.log 2>&1
  python - "$O/stats_$w/run_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:9]:
    print("%-60s calls %5s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
