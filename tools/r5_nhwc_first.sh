#!/bin/bash
# round 5, first GPU contact of the channels-last kernels: parity tests, then the step NCHW vs --channels-last
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r5a"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 500 python -m pytest tests/test_gpu_glue.py -x -q > "$OUT/pytest_glue.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 "$OUT/pytest_glue.log"
[ $rc -ne 0 ] && exit $rc
B="--no-cpu-baseline --no-trainer-loop --no-roofline --steps 30 --warmup 10"
timeout -k 10 200 python bench.py $B > "$OUT/bench_nchw.json" 2> "$OUT/bench_nchw.err" || exit 1
timeout -k 10 300 python bench.py $B --channels-last > "$OUT/bench_nhwc.json" 2> "$OUT/bench_nhwc.err" || exit 1
timeout -k 10 200 python bench.py $B --amp bf16 --graph > "$OUT/bench_nchw_bf16.json" 2> "$OUT/bench_nchw_bf16.err" || exit 1
timeout -k 10 300 python bench.py $B --amp bf16 --graph --channels-last > "$OUT/bench_nhwc_bf16.json" 2> "$OUT/bench_nhwc_bf16.err" || exit 1
python - "$OUT" <<'PY'
import json,sys,os
for n in ("bench_nchw","bench_nhwc","bench_nchw_bf16","bench_nhwc_bf16"):
    d=json.loads([l for l in open(os.path.join(sys.argv[1],n+".json")) if l.startswith("{")][-1])
    print("%-18s %7.1f img/s  %.3f ms" % (n, d["value"], d["ms_per_step"]))
PY
