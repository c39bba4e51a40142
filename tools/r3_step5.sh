#!/bin/bash
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r3e"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
timeout -k 10 200 python bench.py --no-cpu-baseline --no-trainer-loop --steps 40 --warmup 10 > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(round(d["value"],1), {k:d["roofline"][k] for k in ("frac","launch_us")})
PY
timeout -k 10 200 python tools/kbench.py --what train,ident,pre --reps 20 > "$OUT/kbench.txt" 2>&1; tail -5 "$OUT/kbench.txt"
