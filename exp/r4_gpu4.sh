#!/bin/bash
# round 4, call 4: MSM windows / accumulation loop through plain hint stores instead of compiled replays: parity, then the headline
# bench A/B (H2E_NO_PLAIN_STORE=1 = the replays) in one box; hardware counters of the packed and the plain small-batch expansion
cd "$(dirname "$0")/.."
O=gpurun_out/r4_4; mkdir -p $O
timeout 1800 python -m pytest tests/test_parity_gpu.py tests/test_ops_gpu.py tests/test_check_gpu.py -m gpu -x -q -k "not full_size and not batch_64 and not batch_16 and not soak" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
bench() {  # tag [env...] -- args
tag=$1; shift
envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
env "${envs[@]}" timeout 900 python bench.py --sub --suite main --traffic off --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', round(d['single_batch_ms'],3), 'chain', [round(v,2) for v in r['value_chain_ms'] if v > 0.3], 'x', [round(v,2) for v in r['expansion_ms'] if v > 0.3], 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for rep in 1 2; do
bench msm_store_$rep X=1 --
bench msm_replay_$rep H2E_NO_PLAIN_STORE=1 --
done
bench job_store X=1 -- --job-tiles 1024
bench job_replay H2E_NO_PLAIN_STORE=1 -- --job-tiles 1024
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for mode in packed plain; do
if [ $mode = plain ]; then export H2E_TUNE=0,2,0,0,0,1; else unset H2E_TUNE; fi
export H2E_PAIRING_CUT=8
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_$mode -o run --output-format csv -- python3 bench.py --sub --suite main --workload pairing_bn256 --units 8 --ring 1 --steps 2 --warmup 1 --latency-steps 0 --no-cpu-baseline --traffic off > $O/pmc_$mode.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU -d $O/pmc2_$mode -o run --output-format csv -- python3 bench.py --sub --suite main --workload pairing_bn256 --units 8 --ring 1 --steps 2 --warmup 1 --latency-steps 0 --no-cpu-baseline --traffic off > $O/pmc2_$mode.log 2>&1
done
unset H2E_TUNE H2E_PAIRING_CUT
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r4_4/pmc*_*")):
    if not glob.glob(d + "/**/*counter_collection.csv", recursive=True): print(d, "no counters"); continue
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        if "h2e_run_tape" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[(k, row["Counter_Name"])] += 1
    for k, v in agg.items():
        print(d.split("/")[-1], k, {c: round(x / max(1, n[(k, c)])) for c, x in v.items()})
PY
