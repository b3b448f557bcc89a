#!/bin/bash
# round 5, call 17: where a round of the digit chain goes NOW (two passes, pairs, long combinations, division in the loader wave)
cd "$(dirname "$0")/.."
O=gpurun_out/r5_17; mkdir -p $O
( bash exp/wave_stamps.sh && python exp/wave_stamps.py 64 bn256 ) > $O/stamps_bn256.log 2>&1; tail -32 $O/stamps_bn256.log
( FPK=1 bash exp/wave_stamps.sh && python exp/wave_stamps.py 16 bls12_381 ) > $O/stamps_bls.log 2>&1; tail -32 $O/stamps_bls.log
