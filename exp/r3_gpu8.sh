#!/bin/bash
# the driver's own command: python bench.py (whole metric in one line)
cd "$(dirname "$0")/.."
O=gpurun_out/r3h; mkdir -p $O
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3h/bench_default.json").read().strip().splitlines()[-1])
def show(k, x):
    if "error" in x: print(k, "ERROR", x["error"][:300]); return
    print(k, "ms/step %.2f single %.2f value %.3e whole %.3f roof %.3f (%s) cpu %s wall %s" % (x["ms_per_step"], x.get("single_batch_ms") or -1, x["value"], x["whole_step"]["frac"], x["roofline"]["frac"], x["roofline"]["kernel"][:40], (x.get("cpu_baseline") or {}).get("value"), x.get("wall_s")))
show("msm", d)
for k, x in d.get("also", {}).items(): show(k, x)
PY
