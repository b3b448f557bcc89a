#!/bin/bash
# round 4, call 1: (a) the field chain's tail wait under the pairing parity tests, (b) the 2^20-point job in five
# fresh processes after the record / gather path is warmed outside the timed region, (c) batches smaller than a wave: one GPU's
# share of configs[3] / configs[4] at 8 GPUs, pipelined and alone, with kernel stats of the single-batch form
cd "$(dirname "$0")/.."
O=gpurun_out/r4_1; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "pairing" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
for i in 1 2 3 4 5; do
timeout 600 python bench.py --sub --suite main --job-tiles 1024 --traffic off --no-cpu-baseline > $O/job$i.json 2> $O/job$i.err
python -c "
import json; d=json.loads(open('$O/job$i.json').read().strip().splitlines()[-1]); print('job', round(d['ms_per_step'],2), d['summary'])" || tail -3 $O/job$i.err
done
small() {  # workload units ring tag
timeout 600 python bench.py --sub --suite main --workload $1 --units $2 --ring $3 --traffic off --no-cpu-baseline > $O/$4.json 2> $O/$4.err
python -c "
import json; d=json.loads(open('$O/$4.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$4', 'ms/step', round(d['ms_per_step'],3), 'single', round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3), 'x', round(sum(r['expansion_ms']),3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$4.err
}
small pairing_bn256 8 3 bn8_r3
small pairing_bn256 8 1 bn8_r1
small pairing_bn256 64 1 bn64_r1
small pairing_bls12_381 2 3 bls2_r3
small pairing_bls12_381 2 1 bls2_r1
small pairing_bls12_381 16 1 bls16_r1
small pairing_bls12_381 16 3 bls16_r3
small pairing_bls12_381 64 1 bls64_r1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "pairing_bn256 8" "pairing_bls12_381 16" "pairing_bls12_381 2"; do
set -- $cfg
rocprofv3 --kernel-trace --stats -d $O/stats_$1_$2 -o run --output-format csv -- python3 bench.py --sub --suite main --workload $1 --units $2 --ring 1 --latency-steps 0 --steps 10 --no-cpu-baseline --traffic off > $O/stats_$1_$2.log 2>&1
head -12 $O/stats_$1_$2/run_kernel_stats.csv | cut -c1-150
done
