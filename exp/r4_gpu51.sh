#!/bin/bash
# round 4, call 51: counters of the packed expansion with and without the order tables (16 x bls12_381): instructions, busy cycles
cd "$(dirname "$0")/.."
O=gpurun_out/r4_51; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in order tape; do
if [ $v = tape ]; then export H2E_TUNE=0,3,0,0,0,2; else unset H2E_TUNE; fi
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace -d $O/$v -o run --output-format csv -- python3 exp/pmc_packed.py bls12_381 16 > $O/$v.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --kernel-trace -d $O/${v}_b -o run --output-format csv -- python3 exp/pmc_packed.py bls12_381 16 > $O/${v}_b.log 2>&1
done
ls $O/*/ | head -20
