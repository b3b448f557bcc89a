#!/bin/bash
# round 4, call 11: export tests + the consumer-ready step with the flags staged once per block
cd "$(dirname "$0")/.."
O=gpurun_out/r4_11; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_shape_gpu.py -m gpu -x -q -k "export or fixed or columns" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for i in 1 2; do
timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline --workload msm --ring 1 --steps 3 --warmup 1 --latency-steps 0 --consumer-ready 3 > $O/consumer$i.json 2> $O/consumer$i.err
python -c "
import json; d=json.loads(open('$O/consumer$i.json').read().strip().splitlines()[-1]); print('consumer_ready_ms_per_step', d['consumer_ready_ms_per_step'], 'ms_per_step', d['ms_per_step'])" || tail -3 $O/consumer$i.err
done
python exp/next_rows_bench.py > $O/next_rows.json 2> $O/next_rows.err; tail -c 1500 $O/next_rows.json
