#!/bin/bash
# round 4, call 33: kernel traces of the pipelined 16 x bls12_381 step at ring 4 and with the fix-ups on the side streams (H2E_SCHED=5):
# what keeps a deeper ring from paying?
cd "$(dirname "$0")/.."
O=gpurun_out/r4_33; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d $O/ring4 -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --workload pairing_bls12_381 --ring 4 > $O/ring4.log 2>&1
H2E_SCHED=5 rocprofv3 --kernel-trace -d $O/ring3_s5 -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --workload pairing_bls12_381 > $O/ring3_s5.log 2>&1
H2E_SCHED=5 rocprofv3 --kernel-trace -d $O/ring6_s5 -o run --output-format csv -- python3 bench.py --sub --suite main --no-cpu-baseline --traffic off --workload pairing_bls12_381 --ring 6 > $O/ring6_s5.log 2>&1
ls $O/*; tail -c 300 $O/ring4.log
