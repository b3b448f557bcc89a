#!/bin/bash
# digit-parallel field chain: parity first, then the pairing benches (A/B against the lane kernel)
cd "$(dirname "$0")/.."
O=gpurun_out/r3k; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "pairing or digest or ops or tower" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
for w in pairing_bn256 pairing_bls12_381; do
  timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline > $O/$w.json 2> $O/$w.err
  timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline --ring 1 --latency-steps 0 > $O/${w}_ring1.json 2> $O/${w}_ring1.err
  H2E_FIELD_CHAIN=lanes timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline > $O/${w}_lanes.json 2> $O/${w}_lanes.err
done
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("single_batch_ms"), d["roofline"].get("kernel"), d["roofline"]["frac"], d.get("kernels_ms"))
except Exception as e: print("ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
