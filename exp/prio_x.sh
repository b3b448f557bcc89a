#!/bin/bash
# expansion waves at the chain kernels' priority (H2E_TUNE xcache field bit 1) vs default
for t in "0,0,0" "0,2,0" "0,0,0" "0,2,0"; do
  echo -n "tune $t: "
  H2E_TUNE=$t exp/ab_lib.sh default 2>&1 | head -1
done
