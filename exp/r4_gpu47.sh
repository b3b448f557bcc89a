#!/bin/bash
# round 4, call 47: the MSM step is half a run's latency (two buffer sets = two runs in flight; steps alternate 12.95 / 18.3 ms in
# profiles/r4_u_msm): kernel trace of the pipelined step with the run's last small fix-ups on the fix-up stream (H2E_SCHED=132) - what
# the four small segments at the end of a run (4.4 ms) really wait for
cd "$(dirname "$0")/.."
O=gpurun_out/r4_47; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
H2E_SCHED=132 rocprofv3 --kernel-trace -d $O/s132 -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --workload msm --steps 12 --warmup 3 --latency-steps 0 > $O/s132.log 2>&1
tail -c 400 $O/s132.log
for rep in 1 2; do
for s in 4 132; do
H2E_SCHED=$s timeout 600 python bench.py --sub --suite main --no-cpu-baseline --traffic off --workload msm > $O/msm_s${s}_$rep.json 2> $O/msm_s${s}_$rep.err
python -c "
import json; d=json.loads(open('$O/msm_s${s}_$rep.json').read().strip().splitlines()[-1]); print('msm_s${s}_$rep', round(d['ms_per_step'],3), d['single_batch_ms'] and round(d['single_batch_ms'],3))"
done
done
