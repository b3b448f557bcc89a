#!/bin/bash
# round 5, call 2: (a) the depth-balanced field chains: pairing parity tests + A/B against H2E_FIELD_NO_REBALANCE=1 (same library: a
# program-creation knob), ring 1 and pipelined; (b) labelled kernel timeline of the pipelined MSM (which run / segment every kernel of
# three consecutive runs belongs to)
cd "$(dirname "$0")/.."
O=gpurun_out/r5_2; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_pyref_gpu.py -m gpu -x -q -k "pairing" > $O/pytest_pairing.log 2>&1; echo "pytest pairing rc $?"; tail -3 $O/pytest_pairing.log
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2; do
for v in new old; do
  if [ $v = old ]; then export H2E_FIELD_NO_REBALANCE=1; else unset H2E_FIELD_NO_REBALANCE; fi
  for w in pairing_bn256 pairing_bls12_381; do
    timeout 300 $B --workload $w --ring 1 --latency-steps 0 > $O/${w}_ring1_${v}_$rep.json 2> $O/${w}_ring1_${v}_$rep.err
    timeout 300 $B --workload $w > $O/${w}_${v}_$rep.json 2> $O/${w}_${v}_$rep.err
  done
done
done
unset H2E_FIELD_NO_REBALANCE
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_2/pairing_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms"), [round(x, 2) for x in r["value_chain_ms"]], [round(x, 2) for x in r["expansion_ms"]]))
    except Exception as e:
        print(f, "failed", e)
PY
STEPS=8 WARM=2 RUNS=3 bash exp/trace_labelled.sh r5_2/msm_labelled --workload msm
head -5 gpurun_out/r5_2/msm_labelled/labelled.txt
