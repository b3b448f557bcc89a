#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3i; mkdir -p $O
timeout 1800 python -m pytest tests/test_parity_gpu.py -m gpu -q -k "digest" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -8 $O/pytest.log
timeout 900 python bench.py --job-tiles 1024 --traffic off --no-cpu-baseline > $O/job.json 2> $O/job.err
timeout 600 python bench.py --suite main --traffic off --no-cpu-baseline > $O/msm.json 2> $O/msm.err
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("single_batch_ms"), d["whole_step"]["frac"], d["roofline"]["frac"], d["roofline"]["expansion_ms"])
except Exception as e: print("ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
