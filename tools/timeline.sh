#!/bin/bash
# kernel-trace timeline of the replayed step:  gpurun --timeout 600 -- 'bash tools/timeline.sh [tag] [bench args]'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-c1}"; shift
OUT="$ROOT/gpurun_out/timeline_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-trainer-loop --no-roofline --steps 12 --warmup 4 "$@" > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
python3 "$ROOT/tools/timeline.py" "$OUT/raw" --steps 8 > "$OUT/timeline.txt" 2>&1
find "$OUT/raw" -name "*kernel_trace.csv" -delete
cat "$OUT/timeline.txt"
