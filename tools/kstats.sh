#!/bin/bash
# kernel-trace statistics of the bench's step (what profiles/r04_bench_kernel_stats.txt is made from):
#   gpurun --timeout 900 -- 'bash tools/r4_stats.sh [tag] [bench args]'  ->  gpurun_out/stats_<tag>/ + summary on stdout
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-bench}"; shift
OUT="$ROOT/gpurun_out/stats_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/raw" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-trainer-loop --no-roofline --steps 30 --warmup 10 "$@" > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
python3 "$ROOT/tools/summarize_rocprof.py" "$OUT/raw" "$OUT/kernel_stats.txt" "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-trainer-loop --no-roofline --steps 30 --warmup 10 $*   (MI355X, 1 GPU)" > /dev/null
find "$OUT/raw" -name "*kernel_trace.csv" -delete      # (tens of MB; the summary is what is kept)
head -90 "$OUT/kernel_stats.txt" | cut -c1-175
grep '^{' "$OUT/bench.log" | tail -1 | cut -c1-200
