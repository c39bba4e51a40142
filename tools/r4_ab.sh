#!/bin/bash
# A/B of library builds inside the real step (like tools/ab_bench.sh) for several bench configurations:
#   gpurun -- 'CONFIGS="c2 c4 c3" bash tools/r4_ab.sh'      every libmdx_hip.so / libmdx_ab_*.so in the package directory
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
OUT="$ROOT/gpurun_out/ab_r4"; mkdir -p "$OUT"; cd "$ROOT"
run() { tag=$1; n=$2; lib=$3; shift 3; MDX_LIB="$lib" timeout -k 10 300 python bench.py --no-cpu-baseline --no-trainer-loop --steps 30 --warmup 8 "$@" > "$OUT/$tag.$n.json" 2> "$OUT/$tag.$n.err" || { echo "$tag $n FAILED"; tail -3 "$OUT/$tag.$n.err"; return 1; }
python - "$OUT/$tag.$n.json" "$tag" "$n" <<'PY' | tee -a "$OUT/summary.txt"
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("%-4s %-22s %7.1f img/s  kernel %6.1f us  frac %.4f" % (sys.argv[2], sys.argv[3], d["value"], r["launch_us"], r["frac"]))
PY
}
for cfg in ${CONFIGS:-c2}; do
    case $cfg in
        c2) args="" ;;
        c4) args='--frame-ids 0_-1_1_s' ;;
        c3) args="--height 320 --width 1024 --num-layers 50 --batch 8 --amp bf16" ;;
    esac
    for lib in "$PK"/libmdx_hip.so "$PK"/libmdx_ab_*.so; do
        n=$(basename "$lib" .so)
        if [ "$cfg" = c4 ]; then run $cfg "$n" "$lib" --frame-ids "0 -1 1 s" || exit 1; else run $cfg "$n" "$lib" $args || exit 1; fi
    done
done
