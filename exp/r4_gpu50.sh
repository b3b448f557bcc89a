#!/bin/bash
# round 4, call 50: a pipelined run's small fix-ups on a stream of the slot's own (H2E_SCHED=260): the MSM run's last four small segments
# without a fix-up between every two of them and without queueing behind the windows' fix-ups
cd "$(dirname "$0")/.."
O=gpurun_out/r4_50; mkdir -p $O
for rep in 1 2 3; do
for s in 4 260; do
H2E_SCHED=$s timeout 600 python bench.py --sub --suite main --no-cpu-baseline --traffic off --workload msm > $O/msm_s${s}_$rep.json 2> $O/msm_s${s}_$rep.err
python -c "
import json; d=json.loads(open('$O/msm_s${s}_$rep.json').read().strip().splitlines()[-1]); print('msm_s${s}_$rep', round(d['ms_per_step'],3), d['single_batch_ms'] and round(d['single_batch_ms'],3))" || tail -3 $O/msm_s${s}_$rep.err
done
done
for s in 4 260; do
H2E_SCHED=$s timeout 600 python bench.py --sub --suite main --no-cpu-baseline --traffic off --workload pairing_bls12_381 > $O/bls16_s$s.json 2> $O/bls16_s$s.err
python -c "
import json; d=json.loads(open('$O/bls16_s$s.json').read().strip().splitlines()[-1]); print('bls16_s$s', round(d['ms_per_step'],3), d['single_batch_ms'] and round(d['single_batch_ms'],3))"
done
