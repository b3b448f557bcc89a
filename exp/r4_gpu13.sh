#!/bin/bash
# round 4, call 13: the value chain next to a fill / copy kernel instead of an expansion (exp/chain_vs_fill.py)
cd "$(dirname "$0")/.."
O=gpurun_out/r4_13; mkdir -p $O
timeout 600 python exp/chain_vs_fill.py 64 > $O/chain_vs_fill_64.txt 2>&1; cat $O/chain_vs_fill_64.txt | tail -6
timeout 600 python exp/chain_vs_fill.py 8 > $O/chain_vs_fill_8.txt 2>&1; cat $O/chain_vs_fill_8.txt | tail -6
