#!/bin/bash
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/gpu_tests"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 1100 python -m pytest tests -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest_gpu.log"
