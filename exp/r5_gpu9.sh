#!/bin/bash
# round 5, call 9: more launch boundaries in the big expansion (the dispatcher looks at other queues at a launch boundary of a big grid:
# the next run's tiny chain kernels wait for one): H2E_X_PARTS / H2E_X_SPLIT sweep, MSM pipelined
cd "$(dirname "$0")/.."
O=gpurun_out/r5_9; mkdir -p $O
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
run() { # tag, env...
  t=$1; shift
  env "$@" timeout 400 $B --workload msm > $O/msm_$t.json 2> $O/msm_$t.err
}
run p3_a H2E_X_PARTS=3
run p6_s30 H2E_X_PARTS=6 H2E_X_SPLIT=30
run p8_s20 H2E_X_PARTS=8 H2E_X_SPLIT=20
run p5_s25 H2E_X_PARTS=5 H2E_X_SPLIT=25
run p8_s45 H2E_X_PARTS=8 H2E_X_SPLIT=45
run p3_b H2E_X_PARTS=3
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_9/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms") and round(d["single_batch_ms"], 3), [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
