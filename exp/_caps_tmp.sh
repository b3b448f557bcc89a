cd /root/repo
O=gpurun_out/r5_20; mkdir -p $O
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
for cap in 54 66 84 108; do
  export H2E_FIELD_STEP=$cap
  for w in pairing_bn256 pairing_bls12_381; do
    timeout 300 $B --workload $w --ring 1 > $O/${w}_ring1_rows$cap.json 2> $O/${w}_ring1_rows$cap.err
  done
  timeout 300 $B --workload pairing_bn256 > $O/pairing_bn256_rows$cap.json 2> $O/pairing_bn256_rows$cap.err
  timeout 300 $B --workload pairing_bn256 --units 8 > $O/pairing_bn256_share8_rows$cap.json 2> $O/pairing_bn256_share8_rows$cap.err
done
unset H2E_FIELD_STEP
python exp/bench_lines.py $O
