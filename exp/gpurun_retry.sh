#!/bin/bash
# gpurun with retries while no slot / box is free (exit code 3 / "transient"): exp/gpurun_retry.sh <timeout> '<command>'
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
  echo "$out" | tail -60
  if echo "$out" | grep -q "status=transient"; then sleep 120; continue; fi
  break
done
