#!/bin/bash
# round 3: wave-mode level replay vs the four-wave kernels (pairing checks)
cd "$(dirname "$0")/.."
O=gpurun_out/r3b; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for wl in pairing_bn256 pairing_bls12_381; do
  for mode in wave pair; do
    for ring in 1 0; do
      R=""; [ $ring = 1 ] && R="--ring 1"
      H2E_LEVEL_MODE=$mode python bench.py --workload $wl --steps 12 --warmup 3 --traffic off --no-cpu-baseline --latency-steps 0 $R > $O/${wl}_${mode}_ring${ring}.json 2> $O/${wl}_${mode}_ring${ring}.err
    done
  done
done
tail -3 $O/pytest.log
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["value_chain_ms"], d["roofline"]["expansion_ms"])
except Exception as e: print("ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-500:])
PY
done
