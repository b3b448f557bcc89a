#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3v; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_bench_gpu.py -m gpu -x -q -k "digest or job" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for i in 1 2 3; do
timeout 900 python bench.py --sub --suite main --job-tiles 1024 --traffic off --no-cpu-baseline --latency-steps 0 > $O/job$i.json 2> $O/job$i.err
python -c "
import json; d=json.loads(open('$O/job$i.json').read().strip().splitlines()[-1]); r=d['roofline']; print('job', round(d['ms_per_step'],2), 'x', [round(x,1) for x in r['expansion_ms'] if x>0.5], 'chain', [round(x,1) for x in r['value_chain_ms'] if x>0.5])" || tail -3 $O/job$i.err
done
timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0 > $O/plain.json 2> $O/plain.err
python -c "
import json; d=json.loads(open('$O/plain.json').read().strip().splitlines()[-1]); r=d['roofline']; print('plain', round(d['ms_per_step'],2), 'x', [round(x,1) for x in r['expansion_ms'] if x>0.5], 'chain', [round(x,1) for x in r['value_chain_ms'] if x>0.5])"
