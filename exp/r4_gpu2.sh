#!/bin/bash
# round 4, call 2: the packed expansion (batches smaller than a wave) and the device-side constraint check under the whole -m gpu
# suite; the small batches again, packed and - H2E_TUNE's sixth field - through the plain kernel in the same box; the per-instance
# run-size micro-benchmark (consumer-ready columns)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_2; mkdir -p $O
timeout 900 python -m pytest tests/test_check_gpu.py -m gpu -x -q -k "small or needs or wrong" > $O/pytest_check_small.log 2>&1; echo "check small rc $?"; tail -5 $O/pytest_check_small.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_check_gpu.py > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
timeout 1200 python -m pytest tests/test_check_gpu.py -m gpu -x -q -k "full_size" > $O/pytest_check_full.log 2>&1; echo "check full rc $?"; tail -5 $O/pytest_check_full.log
small() {  # workload units ring tag [env]
env $5 timeout 600 python bench.py --sub --suite main --workload $1 --units $2 --ring $3 --traffic off --no-cpu-baseline > $O/$4.json 2> $O/$4.err
python -c "
import json; d=json.loads(open('$O/$4.json').read().strip().splitlines()[-1]); r=d['roofline']; x=r.get('expansion', r)
print('$4', 'ms/step', round(d['ms_per_step'],3), 'single', round(d['single_batch_ms'],3), 'chain', round(sum(r['value_chain_ms']),3), 'x', round(sum(r['expansion_ms']),3), 'x frac', round(x['frac'],3), 'whole', round(d['whole_step']['frac'],3))" || tail -3 $O/$4.err
}
for rep in 1 2; do
small pairing_bn256 8 1 bn8_r1_packed_$rep
small pairing_bn256 8 1 bn8_r1_plain_$rep H2E_TUNE=0,2,0,0,0,1
small pairing_bls12_381 16 1 bls16_r1_packed_$rep
small pairing_bls12_381 16 1 bls16_r1_plain_$rep H2E_TUNE=0,2,0,0,0,1
small pairing_bls12_381 2 1 bls2_r1_packed_$rep
small pairing_bls12_381 2 1 bls2_r1_plain_$rep H2E_TUNE=0,2,0,0,0,1
done
small pairing_bn256 8 3 bn8_r3_packed
small pairing_bls12_381 16 3 bls16_r3_packed
small pairing_bls12_381 2 3 bls2_r3_packed
small pairing_bn256 32 1 bn32_r1_packed
small pairing_bn256 32 1 bn32_r1_plain H2E_TUNE=0,2,0,0,0,1
./exp/ubench/colrun 512 > $O/colrun_512.txt 2>&1; cat $O/colrun_512.txt
./exp/ubench/colrun 64 > $O/colrun_64.txt 2>&1; cat $O/colrun_64.txt
