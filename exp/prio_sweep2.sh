#!/bin/bash
# stream priorities "expand,chain/side,fixup" (lower = higher priority) with the current code
for p in "0,0,0" "0,-1,0" "0,-1,-1" "-1,0,0" "0,0,0" "0,-1,0"; do
  echo -n "prio $p: "
  H2E_STREAM_PRIORITIES=$p exp/ab_lib.sh default 2>&1 | head -1
done
