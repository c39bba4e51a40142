#!/bin/bash
# A/B of kernel builds inside the real step: bench.py's in-step HIP-event timing of the training kernel
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
OUT="$ROOT/gpurun_out/abb"; mkdir -p "$OUT"; cd "$ROOT"
run() { n=$1; lib=$2; shift 2; MDX_LIB="$lib" timeout -k 10 200 python bench.py --no-cpu-baseline --no-trainer-loop --steps 40 --warmup 10 "$@" > "$OUT/$n.json" 2> "$OUT/$n.err"; python - "$OUT/$n.json" "$n" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
    print("%-28s %6.1f img/s  kernel %6.1f us  frac %.4f" % (sys.argv[2], d["value"], r["launch_us"], r["frac"]))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
for lib in "$PK"/libmdx_hip.so "$PK"/libmdx_ab_*.so; do
    n=$(basename "$lib" .so)
    run "$n" "$lib"
    [ -n "$AB_NOPROLOGUE" ] && run "${n}_noprologue" "$lib" --no-prologue
done
