#!/bin/bash
# round 4, call 17: time of the device-side constraint check at BASELINE's batch sizes; the default bench line with the pairings' PMC traffic
cd "$(dirname "$0")/.."
O=gpurun_out/r4_17; mkdir -p $O
timeout 900 python exp/check_time.py > $O/check_time.txt 2>&1; grep -v amdgpu.ids $O/check_time.txt | tail -12
( time python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4_17/bench.json").read().strip().splitlines()[-1])
print(json.dumps(d["summary"]))
for k in ("pairing_bn256", "pairing_bls12_381"):
    x = d["also"][k]["roofline"].get("expansion", d["also"][k]["roofline"])
    print(k, "expansion traffic", x.get("traffic"), "algorithmic", x.get("algorithmic_bytes_per_launch"), x.get("traffic_note"))
PY
