#!/bin/bash
# round 5, call 16: ops per expansion sub-range of the pairing programs (H2E_PAIRING_CUT) with the round's shorter chains: the 64-check
# expansion moves 1.24 x its algorithmic bytes (the MSM's: 1.10 x) - longer sub-ranges re-read fewer operands and store fewer escaping values
cd "$(dirname "$0")/.."
O=gpurun_out/r5_16; mkdir -p $O
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
for cut in 16 24 32 48; do
  export H2E_PAIRING_CUT=$cut
  timeout 300 $B --workload pairing_bn256 > $O/bn256_64_cut$cut.json 2> $O/bn256_64_cut$cut.err
  timeout 300 $B --workload pairing_bn256 --ring 1 > $O/bn256_64_ring1_cut$cut.json 2> $O/bn256_64_ring1_cut$cut.err
  timeout 300 $B --workload pairing_bn256 --units 8 > $O/bn256_share8_cut$cut.json 2> $O/bn256_share8_cut$cut.err
done
for cut in 8 12 16 24; do
  export H2E_PAIRING_CUT=$cut
  timeout 300 $B --workload pairing_bls12_381 > $O/bls16_cut$cut.json 2> $O/bls16_cut$cut.err
  timeout 300 $B --workload pairing_bls12_381 --ring 1 > $O/bls16_ring1_cut$cut.json 2> $O/bls16_ring1_cut$cut.err
done
unset H2E_PAIRING_CUT
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_16/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f chain %s x %s" % (d["ms_per_step"], [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.2]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
