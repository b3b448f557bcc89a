#!/bin/bash
for t in 0 8 1; do
echo "scan test mask $t"
H2E_TUNE=0,0,0,$t exp/trace.sh scanx$t --no-check
grep predict_windows gpurun_out/scanx$t/timeline.txt | cut -c1-100 | tail -6
done
