#!/bin/bash
# counters of the pairings' digit chain (one h2e_run of 64 bn256 / 16 bls12_381 checks): bash exp/pmc_chain.sh -> gpurun_out/pmc_chain/*.txt
cd "$(dirname "$0")/.."
O=gpurun_out/pmc_chain; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for cfg in "bn256 64" "bls12_381 16"; do
  set -- $cfg
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAVES"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $grp -d $O/$1_$tag -o run --output-format csv -- python3 exp/pmc_packed.py $1 $2 > $O/$1_$tag.log 2>&1
  done
done
python - <<'PY'
import csv, glob, collections
for curve in ("bn256", "bls12_381"):
    tot = collections.OrderedDict()
    for f in sorted(glob.glob(f"gpurun_out/pmc_chain/{curve}_*/**/*counter_collection.csv", recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if "h2e_field_chain_digits" in r["Kernel_Name"]]
        # the second run's two launches = the last two dispatches
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-2:]
        for r in rows:
            if int(r["Dispatch_Id"]) in ids:
                tot.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    print(curve)
    for k, v in tot.items():
        v.sort()
        print("  %-24s %s" % (k, "  ".join("%.4g" % x for _, x in v)))
PY
