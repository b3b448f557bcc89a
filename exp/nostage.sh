#!/bin/bash
export H2E_LIB=$PWD/exp/_dbg/libh2e_dbg.so
for ns in 0 1; do
echo "no_stage $ns"
if [ $ns = 1 ]; then export H2E_NO_STAGE=1; fi
exp/trace.sh nostage$ns
grep "h2e_replay" gpurun_out/nostage$ns/timeline.txt | cut -c1-100 | tail -8
done
