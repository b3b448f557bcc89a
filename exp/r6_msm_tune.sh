#!/bin/bash
# MSM step with three runs in flight (h2e_ring): scheduling knobs A/B in one box (each line a fresh process): bash exp/r6_msm_tune.sh <outdir>
O=${1:-gpurun_out/r6_tune}; mkdir -p $O
B="--sub --suite main --traffic off --no-cpu-baseline --latency-steps 0 --full-line --steps 40 --warmup 6"
run() { name=$1; shift; env "$@" python bench.py $B $EXTRA > $O/$name.json 2> $O/$name.err; echo "$name $(grep -o '"ms_per_step": [0-9.]*' $O/$name.json | head -1)"; }
run base_a H2E_NOP=1
run sched12 H2E_SCHED=12
run sched6 H2E_SCHED=6
run sched20 H2E_SCHED=20
run parts2 H2E_X_PARTS=2
run parts4 H2E_X_PARTS=4
run split30 H2E_X_SPLIT=30
run split60 H2E_X_SPLIT=60
run prio_chain "H2E_STREAM_PRIORITIES=0,-1,0"
run prio_x "H2E_STREAM_PRIORITIES=-1,0,0"
EXTRA="--ring 4" run ring4 H2E_NOP=1
run base_b H2E_NOP=1
