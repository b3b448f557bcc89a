#!/bin/bash
cd "$(dirname "$0")/.."
B="python bench.py --sub --suite main --traffic off --no-cpu-baseline --latency-steps 0"
for tune in 0,3 84000,3 120000,3 150000,3; do
  for cfg in "pairing_bn256 8 8" "pairing_bls12_381 2 8" "pairing_bn256 64 4" "pairing_bn256 64 8" "pairing_bls12_381 16 8"; do
    set -- $cfg
    echo "tune $tune $1 x $2 ring $3: $(H2E_TUNE=$tune timeout 300 $B --workload $1 --units $2 --ring $3 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)"
  done
done
