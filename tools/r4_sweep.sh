#!/bin/bash
# Round 4: chunk-schedule sweeps of the training kernel inside the real step for the three BASELINE shapes
# (needs libmdx_ab_dev.so = a -DMDX_DEV_SWITCHES build: MDX_BUILD_NAME=libmdx_ab_dev.so MDX_BUILD_DEFINES=-DMDX_DEV_SWITCHES python build.py).
#   gpurun --timeout 1100 -- 'bash tools/r4_sweep.sh'
# A schedule "r0,f0,r1,f1,r2" cuts every column into floor(f0*H/r0) chunks of r0 rows, floor(f1*H/r1) of r1, the rest in r2s.
# "H,1,H,0,H" is the PERSISTENT form of VERDICT r3 next #4 (one wave walks a whole column: no halo rows inside it),
# "H/2,1,.." / "H/3,1,.." a half / a third of a column per wave.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
OUT="$ROOT/gpurun_out/sweep_r4"; mkdir -p "$OUT"; cd "$ROOT"
sweep() {   # tag "schedules" bench-args...
    tag=$1; scheds=$2; shift 2
    echo "== $tag: python bench.py $*" | tee -a "$OUT/$tag.txt"
    for sch in $scheds; do
        MDX_LIB="$PK/libmdx_ab_dev.so" MDX_TRAIN_SCHEDULE="$sch" timeout -k 10 300 python bench.py --no-cpu-baseline --no-trainer-loop --steps 30 --warmup 8 "$@" > "$OUT/s.json" 2> "$OUT/s.err" || { echo "$sch FAILED"; tail -3 "$OUT/s.err"; return 1; }
        python - "$OUT/s.json" "$sch" <<'PY' | tee -a "$OUT/$tag.txt"
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("%-24s kernel %6.1f us  frac %.4f   %7.1f img/s" % (sys.argv[2], r["launch_us"], r["frac"], d["value"]))
PY
    done
}
C2="${C2:-40,0.63,24,0.25,12 192,1,192,0,192 96,1,96,0,96 64,1,64,0,64 48,1,48,0,48 64,0.67,32,0.17,16 48,0.75,24,0,12 56,0.6,28,0.15,14 48,0.5,24,0.25,12 44,0.7,30,0.16,13}"
C4="${C4:-40,0.63,24,0.25,12 64,0.67,32,0.17,16 48,0.75,24,0,12 56,0.6,28,0.15,14 48,0.5,24,0.25,12 32,0.67,16,0.17,8 24,0.63,16,0.25,8}"
C3="${C3:-40,0.6,20,0.25,10 64,0.6,32,0.2,16 80,0.5,40,0.25,20 64,0.8,32,0.1,16 80,0.75,40,0.125,20 40,0.625,24,0.225,12 107,1,107,0,107 160,1,160,0,160 320,1,320,0,320}"
[ -n "$SKIP_C2" ] || sweep c2 "$C2" || exit 1
[ -n "$SKIP_C4" ] || sweep c4 "$C4" --frame-ids "0 -1 1 s" || exit 1
[ -n "$SKIP_C3" ] || sweep c3 "$C3" --height 320 --width 1024 --num-layers 50 --batch 8 --amp bf16 || exit 1
