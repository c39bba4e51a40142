#!/bin/bash
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/imgproc_round3"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_imgproc.py -x -q > "$OUT/pytest_img.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_img.log"
bash tools/trace_imgbench.sh s0 --n 32 --out 192x640
bash tools/trace_imgbench.sh s1 --n 12 --out 96x320
bash tools/trace_imgbench.sh s2 --n 12 --out 48x160
bash tools/trace_imgbench.sh s3 --n 12 --out 24x80
timeout -k 10 200 python tools/loader_cost.py --samples 12 --reps 20 > "$OUT/loader_cost.txt" 2>&1; tail -6 "$OUT/loader_cost.txt"
