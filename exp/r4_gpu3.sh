#!/bin/bash
# round 4, call 3: the two fixes of call 2 (at most 32 groups per packed wave; the checker tests' restore) under the tests that failed,
# then ops per expansion sub-range (H2E_PAIRING_CUT, record time) x packed / plain for the small and the full pairing batches
cd "$(dirname "$0")/.."
O=gpurun_out/r4_3; mkdir -p $O
timeout 1500 python -m pytest tests/test_check_gpu.py tests/test_parity_gpu.py -m gpu -x -q -k "check or value_chain_does_not or smoke or edge" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
small() {  # workload units ring tag [env...]
tag=$4; w=$1; u=$2; rg=$3; shift 4
env "$@" timeout 600 python bench.py --sub --suite main --workload $w --units $u --ring $rg --traffic off --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
python -c "
import json; d=json.loads(open('$O/$tag.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$tag', 'ms/step', round(d['ms_per_step'],3), 'single', round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3), 'x', round(sum(r['expansion_ms']),3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$tag.err
}
for cut in 16 8 6 4; do
small pairing_bn256 8 1 bn8_cut${cut}_packed H2E_PAIRING_CUT=$cut
small pairing_bn256 8 1 bn8_cut${cut}_plain H2E_PAIRING_CUT=$cut H2E_TUNE=0,2,0,0,0,1
small pairing_bls12_381 16 1 bls16_cut${cut}_packed H2E_PAIRING_CUT=$cut
small pairing_bls12_381 16 1 bls16_cut${cut}_plain H2E_PAIRING_CUT=$cut H2E_TUNE=0,2,0,0,0,1
small pairing_bls12_381 2 1 bls2_cut${cut}_packed H2E_PAIRING_CUT=$cut
small pairing_bn256 64 1 bn64_cut${cut} H2E_PAIRING_CUT=$cut
done
small pairing_bn256 64 3 bn64_r3_cut16 H2E_PAIRING_CUT=16
small pairing_bn256 64 3 bn64_r3_cut8 H2E_PAIRING_CUT=8
small pairing_bls12_381 16 3 bls16_r3_cut8 H2E_PAIRING_CUT=8
small pairing_bls12_381 16 3 bls16_r3_cut4 H2E_PAIRING_CUT=4
small pairing_bn256 8 3 bn8_r3_cut4 H2E_PAIRING_CUT=4
