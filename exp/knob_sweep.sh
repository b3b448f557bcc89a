#!/bin/bash
# cheap knob sweep on the current code: hold-back, split percentage, small-expansion threshold
for cfg in "H2E_SCHED=4" "H2E_SCHED=20" "H2E_SCHED=12" "H2E_X_SPLIT=30" "H2E_X_SPLIT=60" "H2E_X_SPLIT=0" "H2E_SCHED=4"; do
  echo -n "$cfg: "
  env $cfg exp/ab_lib.sh default 2>&1 | head -1
done
