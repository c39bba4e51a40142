#!/bin/bash
# chunk-schedule sweep of the training kernel inside the real step (needs a -DMDX_DEV_SWITCHES build as libmdx_ab_dev.so)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
SCHEDULES="${SCHEDULES:-40,0.6,20,0.25,10 40,0.42,24,0.37,12}"
OUT="$ROOT/gpurun_out/sweep"; mkdir -p "$OUT"; cd "$ROOT"
for sch in $SCHEDULES; do
    MDX_LIB="$PK/libmdx_ab_dev.so" MDX_TRAIN_SCHEDULE="$sch" timeout -k 10 200 python bench.py --no-cpu-baseline --no-trainer-loop --steps 40 --warmup 10 > "$OUT/s.json" 2> "$OUT/s.err"
    python - "$OUT/s.json" "$sch" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
    print("%-22s kernel %6.1f us  frac %.4f" % (sys.argv[2], r["launch_us"], r["frac"]))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
done
