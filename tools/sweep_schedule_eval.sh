#!/bin/bash
# chunk-schedule sweep of the FORWARD-ONLY form (validation step) on bench.py's batch (needs libmdx_ab_dev.so, -DMDX_DEV_SWITCHES)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
SCHEDULES="${SCHEDULES:-40,0.63,24,0.25,12 32,0.5,16,0.33,8 24,0.5,12,0.3,6 48,0.5,24,0.25,12 64,0.33,32,0.33,16 20,0.5,10,0.3,5}"
cd "$ROOT"
for sch in $SCHEDULES; do
    echo -n "$sch  "; MDX_LIB="$PK/libmdx_ab_dev.so" MDX_TRAIN_SCHEDULE="$sch" timeout -k 10 200 python tools/valid_step_probe.py 2>&1 | grep "validation step" | sed 's/.*launches)://'
done
