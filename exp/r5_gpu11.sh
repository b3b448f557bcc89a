#!/bin/bash
# round 5, call 11: runs in flight for the small pairing batches now that the chains are shorter (round 4: four; a fifth slowed every chain)
cd "$(dirname "$0")/.."
O=gpurun_out/r5_11; mkdir -p $O
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
for ring in 3 4 5 6 8; do
  timeout 300 $B --workload pairing_bn256 --units 8 --ring $ring > $O/bn256_share8_ring$ring.json 2> $O/bn256_share8_ring$ring.err
  timeout 300 $B --workload pairing_bls12_381 --units 2 --ring $ring > $O/bls_share8_ring$ring.json 2> $O/bls_share8_ring$ring.err
  timeout 300 $B --workload pairing_bls12_381 --ring $ring > $O/bls16_ring$ring.json 2> $O/bls16_ring$ring.err
  timeout 300 $B --workload pairing_bn256 --ring $ring > $O/bn256_64_ring$ring.json 2> $O/bn256_64_ring$ring.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_11/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f chain %s x %s" % (d["ms_per_step"], [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
