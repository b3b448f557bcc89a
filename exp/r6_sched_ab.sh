#!/bin/bash
# MSM step, three runs in flight: the candidates' held-back expansion on the shared expansion stream (H2E_SCHED=4, default until round 6)
# or beside it on the small-expansion stream (12) - alternating in one box
O=${1:-gpurun_out/r6_sched}; mkdir -p $O
B="--sub --suite main --traffic off --no-cpu-baseline --latency-steps 0 --full-line --steps 40 --warmup 6"
for i in 1 2 3; do for sc in 4 12; do
  H2E_SCHED=$sc python bench.py $B > $O/s${sc}_$i.json 2> $O/s${sc}_$i.err; echo "sched $sc run $i $(grep -o '"ms_per_step": [0-9.]*' $O/s${sc}_$i.json | head -1)"
done; done
