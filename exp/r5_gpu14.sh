#!/bin/bash
# round 5, call 14: kernel stats of the pipelined MSM with the tail's phase 1 in digit rows / as one lane per instance
cd "$(dirname "$0")/.."
O=gpurun_out/r5_14; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for v in rows lanes; do
  if [ $v = lanes ]; then export H2E_TUNE=0,3,0,16; else unset H2E_TUNE; fi
  rocprofv3 --kernel-trace --stats -d $O/stats_$v -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --workload msm --steps 10 --warmup 2 --latency-steps 0 > $O/stats_$v.log 2>&1
  python - "$O/stats_$v/run_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("%-56s calls %5s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
done
