#!/bin/bash
# where the column-emission expansion's time goes (wrong results on purpose: --no-check): H2E_COLS_DBG 1 = no working-copy stores,
# 2 = no column stores, 4 = no zero fill of passed rows
O=${1:-gpurun_out/r6_cols}; mkdir -p $O
B="--sub --suite main --workload msm --ring 1 --steps 2 --warmup 1 --latency-steps 0 --consumer-ready 2 --no-cpu-baseline --traffic off --full-line --no-check"
for d in ${DBGS:-0 1 2 3 4 7}; do
  H2E_COLS_DBG=$d python bench.py $B > $O/dbg_$d.json 2> $O/dbg_$d.err
  python - $O/dbg_$d.json $d <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1])
fp=d["consumer_ready_first_pass"]
lm=fp.get("launch_ms",[[]])[-1]
print("dbg", sys.argv[2], "first pass ms", fp.get("ms_per_step"), "windows", lm[8] if len(lm)>8 else None, "cand", lm[5] if len(lm)>5 else None, fp.get("error"))
P
done
