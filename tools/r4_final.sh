#!/bin/bash
# end-of-round check on the GPU box: whole GPU suite, smoke(), the default bench line, kernel statistics of the step and of the
# image-preparation stage.   gpurun --timeout 1200 -- 'bash tools/r4_final.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/final"; mkdir -p "$OUT"; cd "$ROOT"
bash tools/gpu_tests.sh || exit 1
grep -q " passed" gpurun_out/gpu_tests/pytest_gpu.log && ! grep -q " failed" gpurun_out/gpu_tests/pytest_gpu.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > "$OUT/smoke.log" 2>&1 || { tail -5 "$OUT/smoke.log"; exit 1; }
tail -1 "$OUT/smoke.log"
timeout -k 10 600 python bench.py > "$OUT/bench_full.json" 2> "$OUT/bench_full.err" || { tail -5 "$OUT/bench_full.err"; exit 1; }
tail -c 1500 "$OUT/bench_full.json"; echo
bash tools/r4_stats.sh final || exit 1
bash tools/profile_imgproc.sh > "$OUT/profile_imgproc.log" 2>&1 || exit 1
grep "mdx::" "$OUT/profile_imgproc.log" | cut -c1-120
