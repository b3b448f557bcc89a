#!/bin/bash
# shader clock while the pairing bench runs one batch at a time (ring 1) and pipelined (ring 3): is the chain's slowdown under load a clock effect?
cd "$(dirname "$0")/.."
O=gpurun_out/r3_clk; mkdir -p $O
for r in 1 3; do
( for i in $(seq 1 60); do /opt/rocm/bin/rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; sleep 0.25; done ) > $O/clk_ring$r.txt &
S=$!
timeout 600 python bench.py --sub --suite main --workload pairing_bn256 --traffic off --no-cpu-baseline --ring $r --latency-steps 0 --steps 2000 > $O/bn_ring$r.json 2> $O/bn_ring$r.err
kill $S 2>/dev/null; wait $S 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bn_ring$r.json').read().strip().splitlines()[-1]); print('ring $r', round(d['ms_per_step'],3), d['roofline']['value_chain_ms'])"
sort $O/clk_ring$r.txt | uniq -c | sort -rn | head -6
done
/opt/rocm/bin/rocm-smi --showpower 2>/dev/null | head -8
