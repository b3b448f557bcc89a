#!/bin/bash
# round 4, call 35: host time per pipelined submission for the small pairing batches
cd "$(dirname "$0")/.."
O=gpurun_out/r4_35; mkdir -p $O
for cfg in "bls12_381 16 3 1" "bls12_381 16 6 1" "bls12_381 16 3 0" "bls12_381 16 6 0" "bls12_381 2 3 1" "bls12_381 2 6 0" "bn256 8 3 1" "bn256 8 6 0" "bn256 64 3 1"; do
timeout 300 python exp/submit_host_time.py $cfg 2>&1 | tail -1
done | tee $O/host_time.txt
