#!/bin/bash
# round 3 checkpoint: the whole -m gpu suite + headline numbers
cd "$(dirname "$0")/.."
O=gpurun_out/r3n; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -12 $O/pytest.log
timeout 600 python bench.py --suite main --traffic off --no-cpu-baseline > $O/msm.json 2> $O/msm.err
for wl in pairing_bn256 pairing_bls12_381; do
  timeout 600 python bench.py --workload $wl --traffic off --no-cpu-baseline > $O/${wl}.json 2> $O/${wl}.err
done
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("single_batch_ms"), d["roofline"]["value_chain_ms"], d["roofline"]["expansion_ms"])
except Exception as e: print("ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
