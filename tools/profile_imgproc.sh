#!/bin/bash
# rocprofv3 kernel trace of the image-preparation stage (tools/loader_cost.py).  On the GPU box:
#     gpurun --timeout 600 -- 'bash tools/profile_imgproc.sh'
# then here:   bash tools/profile_imgproc.sh --collect r02
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/profile_imgproc"
if [ "$1" = "--collect" ]; then
    tag="${2:-r02}"
    cd "$ROOT" || exit 1
    python tools/summarize_rocprof.py "$OUT/trace" "profiles/${tag}_imgproc_kernel_stats.txt" \
        "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/loader_cost.py --samples 2 --reps 20   (batch 12, 3 frames of 1242x375 -> 192x640 pyramid + jitter)" > /dev/null
    cp "$OUT/loader_cost.json" "profiles/${tag}_loader_cost.json"
    exit 0
fi
mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/loader_cost.py" --workers 12 > "$OUT/loader_cost.json" 2> "$OUT/loader_cost.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/loader_cost.py" --samples 2 --reps 20 > "$OUT/trace.log" 2>&1 || exit 1
cat "$OUT/loader_cost.json"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
head -12 "$f" | cut -c1-150
