#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python bench.py --digest --steps 16 --warmup 4 --traffic off --no-cpu-baseline --latency-steps 0 > $O/digest16.json 2> $O/digest16.err
timeout 900 python bench.py --steps 16 --warmup 4 --traffic off --no-cpu-baseline --latency-steps 0 --units 64 > $O/plain16.json 2> $O/plain16.err
timeout 900 python bench.py --job-tiles 1024 --traffic off --no-cpu-baseline --latency-steps 0 > $O/job.json 2> $O/job.err
timeout 900 python bench.py --job-tiles 2048 --traffic off --no-cpu-baseline --latency-steps 0 > $O/job2048.json 2> $O/job2048.err
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["steps"], d["whole_step"]["frac"], d["roofline"]["frac"])
except Exception as e: print("ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
