#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r3g"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_driver.py -q -k "trainer" > "$OUT/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$OUT/pytest.log"; grep -c "AccumulateGrad" "$OUT/pytest.log"
MDX_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 2 --steps 10 --warmup 3 --one-loop --no-cpu-baseline > "$OUT/bench_gloo2.json" 2> "$OUT/bench_gloo2.err"; echo "gloo2 rc=$?"
python - "$OUT/bench_gloo2.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print({k:d.get(k) for k in ("value","n_gpus","ranks_verified","hip_graph","gradient_exchange")}); print("trainer_loop", d.get("trainer_loop"))
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-2500:])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
