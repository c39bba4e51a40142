#!/bin/bash
# BASELINE.json configs[2..4] on ONE GPU (their 8-GPU form is the driver's to run): bench lines under gpurun_out/configs_r5/
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/configs_r5"; mkdir -p "$OUT"; cd "$ROOT"
run() { n=$1; shift; timeout -k 10 500 python bench.py --no-cpu-baseline --no-trainer-loop --steps 40 --warmup 10 "$@" > "$OUT/$n.json" 2> "$OUT/$n.err" || { echo "$n FAILED"; tail -3 "$OUT/$n.err"; return; }
python - "$OUT/$n.json" "$n" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("%-22s %7.1f img/s  %6.2f ms/step  kernel %s  %6.1f us  frac %.4f" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel"][:44], r["launch_us"], r["frac"]))
PY
}
run c2_bf16_graph --amp bf16 --graph
run c3_r50_320x1024_bf16 --num-layers 50 --height 320 --width 1024 --batch 8 --amp bf16 --graph
run c3_r50_320x1024_fp32 --num-layers 50 --height 320 --width 1024 --batch 8
run c4_mono_stereo_fp32 --frame-ids "0 -1 1 s"
run c4_mono_stereo_bf16 --frame-ids "0 -1 1 s" --amp bf16 --graph
