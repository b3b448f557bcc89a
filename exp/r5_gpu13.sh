#!/bin/bash
# round 5, call 13: phase 1 of the MSM tail's scan in digit rows (h2e_predict_tail_rows): the MSM parity tests (scan fallbacks, general
# scalars = 12-digit field, full size), then the pipelined step against the lane form (H2E_TUNE fourth field bit 3), alternating
cd "$(dirname "$0")/.."
O=gpurun_out/r5_13; mkdir -p $O
timeout 2400 python -m pytest tests/test_parity_gpu.py tests/test_pyref_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "msm or ops" > $O/pytest_msm.log 2>&1; echo "pytest msm rc $?"; tail -4 $O/pytest_msm.log
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2; do
  timeout 400 $B --workload msm > $O/msm_rows_$rep.json 2> $O/msm_rows_$rep.err
  H2E_TUNE=0,3,0,16 timeout 400 $B --workload msm > $O/msm_lanes_$rep.json 2> $O/msm_lanes_$rep.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_13/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms") and round(d["single_batch_ms"], 3), [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
