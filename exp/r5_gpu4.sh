#!/bin/bash
# round 5, call 4: division by the loader wave + two passes per round: the pairing parity / variant / digit-row / check tests, then the
# pairing bench lines (ring 1 = the chain alone, pipelined, the 8-GPU shares)
cd "$(dirname "$0")/.."
O=gpurun_out/${OUT:-r5_4}; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_pyref_gpu.py tests/test_digit_rows_gpu.py tests/test_check_gpu.py -m gpu -x -q -k "pairing or digit or check" > $O/pytest_pairing.log 2>&1; echo "pytest pairing rc $?"; tail -3 $O/pytest_pairing.log
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2; do
  for w in pairing_bn256 pairing_bls12_381; do
    timeout 300 $B --workload $w --ring 1 --latency-steps 0 > $O/${w}_ring1_$rep.json 2> $O/${w}_ring1_$rep.err
    timeout 300 $B --workload $w > $O/${w}_$rep.json 2> $O/${w}_$rep.err
  done
done
timeout 300 $B --workload pairing_bn256 --units 8 > $O/pairing_bn256_share8.json 2> $O/pairing_bn256_share8.err
timeout 300 $B --workload pairing_bls12_381 --units 2 > $O/pairing_bls12_381_share8.json 2> $O/pairing_bls12_381_share8.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/" + __import__("os").environ.get("OUT", "r5_4") + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms") and round(d["single_batch_ms"], 3), [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
