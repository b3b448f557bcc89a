#!/bin/bash
# round 3, first GPU pass: class-round scheduler of the level-parallel replay vs level rounds (pairing checks)
cd "$(dirname "$0")/.."
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing or unsafe" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
for wl in pairing_bn256 pairing_bls12_381; do
  for sched in classes levels; do
    for ring in 1 0; do
      R=""; [ $ring = 1 ] && R="--ring 1"
      S=""; [ $sched = levels ] && S="levels"
      H2E_LEVEL_SCHED=$S python bench.py --workload $wl --steps 12 --warmup 3 --traffic off --no-cpu-baseline $R > $O/${wl}_${sched}_ring${ring}.json 2> $O/${wl}_${sched}_ring${ring}.err
    done
  done
done
tail -3 $O/pytest.log
for f in $O/*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["value_chain_ms"], d["roofline"]["expansion_ms"])
except Exception as e: print("ERR", e)
PY
done
