#!/bin/bash
# round 5, call 8: (a) the checker with its own patch values + the traffic test with its skip reason; (b) MSM pipelined step with the big
# expansion in 2 / 3 / 4 launches (H2E_X_PARTS: the last inverse fix-up, which nothing runs under, shrinks with the last part)
cd "$(dirname "$0")/.."
O=gpurun_out/r5_8; mkdir -p $O
timeout 1500 python -m pytest tests/test_check_gpu.py "tests/test_bench_gpu.py::test_bench_traffic_counters_by_launch_index" tests/test_parity_gpu.py -m gpu -x -q -rs -k "check or traffic or split or msm_tile_full" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2; do
for parts in 2 3 4; do
  H2E_X_PARTS=$parts timeout 400 $B --workload msm > $O/msm_parts${parts}_$rep.json 2> $O/msm_parts${parts}_$rep.err
done
done
H2E_X_PARTS=3 H2E_X_SPLIT=34 timeout 400 $B --workload msm > $O/msm_parts3_split34.json 2> $O/msm_parts3_split34.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_8/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms") and round(d["single_batch_ms"], 3), [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
