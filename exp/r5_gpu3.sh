#!/bin/bash
# round 5, call 3: (a) rows per round of the depth-balanced field chains (H2E_FIELD_STEP: more than one pass of 60 rows per round), with the
# division by division steps in one lane; every bench run asserts status 0 for every unit (a wrong hint is H2E_STATUS_ARITH);
# (b) MSM pipelined step under stream priorities (value-chain streams high)
cd "$(dirname "$0")/.."
O=gpurun_out/r5_3; mkdir -p $O
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for st in 54 80 108 160; do
  export H2E_FIELD_STEP=$st
  for w in pairing_bn256 pairing_bls12_381; do
    timeout 300 $B --workload $w --ring 1 --latency-steps 0 > $O/${w}_ring1_step$st.json 2> $O/${w}_ring1_step$st.err
    timeout 300 $B --workload $w > $O/${w}_step$st.json 2> $O/${w}_step$st.err
  done
done
export H2E_FIELD_STEP=108
timeout 300 $B --workload pairing_bn256 --units 8 > $O/pairing_bn256_share8_step108.json 2> $O/pairing_bn256_share8_step108.err
timeout 300 $B --workload pairing_bls12_381 --units 2 > $O/pairing_bls12_381_share8_step108.json 2> $O/pairing_bls12_381_share8_step108.err
unset H2E_FIELD_STEP
for p in "0,0,0" "0,-1,0" "0,-1,-1" "-1,0,0"; do
  export H2E_STREAM_PRIORITIES=$p
  tag=$(echo $p | tr ',-' '_m')
  timeout 400 $B --workload msm > $O/msm_prio_$tag.json 2> $O/msm_prio_$tag.err
done
unset H2E_STREAM_PRIORITIES
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_3/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms") and round(d["single_batch_ms"], 3), [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
