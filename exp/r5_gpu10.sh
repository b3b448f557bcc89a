#!/bin/bash
# round 5, call 10: does the digit chain's register allocation (120 VGPRs with the division-steps function in the loader wave's path, 57
# with the exponentiation in the rows) cost the PIPELINED pairing steps, where expansion waves share the chains' CUs?  A/B of two builds
# of the bn256 unit in one box, alternating
cd "$(dirname "$0")/.."
O=gpurun_out/r5_10; mkdir -p $O
bash exp/ab_build.sh fermat 0 -DH2E_EXP_FERMAT_DIV > $O/build.log 2>&1 || { tail -5 $O/build.log; exit 1; }
cp halo2ecc_s_amd/libh2e.so $O/shipped.so
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline"
for rep in 1 2 3; do
  for v in shipped fermat; do
    if [ $v = fermat ]; then cp exp/_dbg/libh2e_fermat.so halo2ecc_s_amd/libh2e.so; else cp $O/shipped.so halo2ecc_s_amd/libh2e.so; fi
    timeout 300 $B --workload pairing_bn256 > $O/bn256_${v}_$rep.json 2> $O/bn256_${v}_$rep.err
    timeout 300 $B --workload pairing_bn256 --units 8 > $O/bn256_share8_${v}_$rep.json 2> $O/bn256_share8_${v}_$rep.err
  done
done
cp $O/shipped.so halo2ecc_s_amd/libh2e.so; rm $O/shipped.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5_10/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d["roofline"]
        print(f.split("/")[-1][:-5], "ms/step %.3f single %s chain %s x %s" % (d["ms_per_step"], d.get("single_batch_ms") and round(d["single_batch_ms"], 3), [round(x, 2) for x in r["value_chain_ms"] if x > 0.3], [round(x, 2) for x in r["expansion_ms"] if x > 0.3]))
    except Exception as e:
        print(f, "failed", e, open(f[:-5] + ".err").read()[-300:])
PY
