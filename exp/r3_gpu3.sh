#!/bin/bash
# round 3: after the address-space fix (no FLAT accesses): stamps, pairing parity + timing, MSM headline quick check
cd "$(dirname "$0")/.."
O=gpurun_out/r3c; mkdir -p $O
python exp/wave_stamps.py 64 > $O/stamps.txt 2>&1
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing or int_mul or integer_chip or msm_tile" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for wl in pairing_bn256 pairing_bls12_381; do
  for mode in wave pair; do
    for ring in 1 0; do
      R=""; [ $ring = 1 ] && R="--ring 1"
      H2E_LEVEL_MODE=$mode python bench.py --workload $wl --steps 12 --warmup 3 --traffic off --no-cpu-baseline --latency-steps 0 $R > $O/${wl}_${mode}_ring${ring}.json 2> $O/${wl}_${mode}_ring${ring}.err
    done
  done
done
python bench.py --suite main --traffic off --no-cpu-baseline > $O/msm.json 2> $O/msm.err
cat $O/stamps.txt | tail -8
tail -3 $O/pytest.log
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("single_batch_ms"), d["roofline"]["value_chain_ms"], d["roofline"]["expansion_ms"])
except Exception as e: print("ERR", e, open(sys.argv[1].replace('.json','.err')).read()[-800:])
PY
done
