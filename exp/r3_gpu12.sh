#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r3l; mkdir -p $O
python exp/wave_stamps.py > $O/stamps.txt 2>&1; cat $O/stamps.txt | tail -9
