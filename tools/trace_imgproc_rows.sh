#!/bin/bash
# the image-preparation kernels per scale (rocprofv3 kernel trace of tools/imgbench.py) and per batch (tools/profile_imgproc.sh)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
OUT="$ROOT/gpurun_out/imgproc_rows"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_imgproc.py -x -q > "$OUT/pytest_img.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 "$OUT/pytest_img.log"
[ $rc -eq 0 ] || exit 1
for cfg in "s0 32 192x640" "s1 12 96x320" "s2 12 48x160" "s3 12 24x80"; do
    set -- $cfg
    bash tools/trace_imgbench.sh rows_$1 --n $2 --out $3 | grep resample || exit 1
done
bash tools/profile_imgproc.sh | tail -8
