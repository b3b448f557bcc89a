#!/bin/bash
# run-to-run spread of the three headline lines on one box (five fresh processes each)
cd "$(dirname "$0")/.."
O=gpurun_out/r3_spread; mkdir -p $O
for w in msm pairing_bn256 pairing_bls12_381; do
for i in 1 2 3 4 5; do
timeout 600 python bench.py --sub --suite main --workload $w --traffic off --no-cpu-baseline > $O/${w}_$i.json 2> $O/${w}_$i.err
done
python - "$O" "$w" <<'PY'
import json,sys,glob
o,w=sys.argv[1:3]
ms=[];sb=[]
for f in sorted(glob.glob(f"{o}/{w}_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); ms.append(round(d["ms_per_step"],2)); sb.append(round(d["single_batch_ms"],2))
print(w, "ms_per_step", ms, "single_batch_ms", sb)
PY
done
