#!/bin/bash
# instruction-fetch and issue counters of the expansion kernels (h2e_run_tape*), one h2e_run each of 64 MSM tiles, 64 bn256 and 16 bls12_381
# checks: bash exp/pmc_expansion.sh -> gpurun_out/pmc_expansion/summary.txt
cd "$(dirname "$0")/.."
O=gpurun_out/pmc_expansion; mkdir -p $O
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp && cd $ROOT
for cfg in "msm 64" "bn256 64" "bls12_381 16"; do
  set -- $cfg
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
    tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 600 rocprofv3 --pmc $grp -d $O/$1_$tag -o run --output-format csv -- python3 exp/pmc_run.py $1 $2 > $O/$1_$tag.log 2>&1
  done
done
python - <<'PY' | tee gpurun_out/pmc_expansion/summary.txt
import csv, glob, collections
for what in ("msm", "bn256", "bls12_381"):
    tot = collections.OrderedDict()
    for f in sorted(glob.glob(f"gpurun_out/pmc_expansion/{what}_*/**/*counter_collection.csv", recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if "h2e_run_tape" in r["Kernel_Name"]]
        if not rows:
            continue
        # the second run's launches: the last half of the dispatches; per launch: grid, counter
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})
        ids = ids[len(ids) // 2:]
        for r in rows:
            if int(r["Dispatch_Id"]) in ids:
                tot.setdefault(r["Counter_Name"], []).append((int(r["Dispatch_Id"]), int(r["Grid_Size"]), float(r["Counter_Value"])))
    print(what)
    first = True
    for k, v in tot.items():
        v.sort()
        if first:
            print("  %-28s %s" % ("grid (threads)", "  ".join("%10d" % g for _, g, _ in v)))
            first = False
        print("  %-28s %s" % (k, "  ".join("%10.4g" % x for _, _, x in v)))
PY
