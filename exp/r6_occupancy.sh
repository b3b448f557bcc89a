#!/bin/bash
# what the expansion loses with fewer waves per CU (LDS a column-staging emission would need): H2E_TUNE's xpad = extra dynamic LDS per
# workgroup of the big launches.  19 KB now (8 waves per CU); +7000 -> 6 waves, +12000 -> 5, +20000 -> 4, +33000 -> 3, +60000 -> 2
O=${1:-gpurun_out/r6_occ}; mkdir -p $O
B="--sub --suite main --traffic off --no-cpu-baseline --full-line --steps 6 --warmup 2 --ring 1 --latency-steps 3"
for pad in 0 7000 12000 20000 33000 60000; do
  H2E_TUNE="0,3,$pad,0,0,0" python bench.py $B > $O/pad_$pad.json 2> $O/pad_$pad.err
  python - $O/pad_$pad.json $pad <<'P'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1])
x=d['roofline'].get('expansion', d['roofline'])
print('xpad', sys.argv[2], 'ms/step (h2e_run)', round(d['ms_per_step'],3), 'single', round(d['single_batch_ms'],3), 'window launch ms', round(x['launch_ms'],3), 'frac', round(x['frac'],3))
P
done
