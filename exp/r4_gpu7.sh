#!/bin/bash
# round 4, call 7: the whole -m gpu suite on the round's code (packed expansion, plain hint stores, two-launch pairings, checker,
# no-select operator context, strong-scaling bench), then the default bench line
cd "$(dirname "$0")/.."
O=gpurun_out/r4_7; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; cat $O/bench.time | tail -3
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_7/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"], indent=0))
print("roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
