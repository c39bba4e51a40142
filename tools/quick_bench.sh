#!/bin/bash
# in-step timing of the training kernel + its parity tests:  gpurun -- 'bash tools/quick_bench.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/quick"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_prologue.py tests/test_gpu_golden_r2.py -x -q > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline --no-trainer-loop --steps 40 --warmup 10 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]
print("%6.1f img/s  kernel %6.1f us  frac %.4f" % (d["value"], r["launch_us"], r["frac"]))
PY
done
