#!/bin/bash
# round 3: field chain + hint store for the pairings: parity, then timing
cd "$(dirname "$0")/.."
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
for wl in pairing_bn256 pairing_bls12_381; do
  for ring in 1 0; do
    R=""; [ $ring = 1 ] && R="--ring 1"
    timeout 600 python bench.py --workload $wl --steps 12 --warmup 3 --traffic off --no-cpu-baseline --latency-steps 0 $R > $O/${wl}_ring${ring}.json 2> $O/${wl}_ring${ring}.err
  done
done
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("single_batch_ms"), d["roofline"]["value_chain_ms"], d["roofline"]["expansion_ms"])
except Exception as e: print("ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
