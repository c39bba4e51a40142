#!/bin/bash
# round-3 first GPU pass: the data-parallel step on one GPU (RCCL group of one), graph trajectory, diagnostics
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r3a"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_driver.py -x -q -k "trainer" > "$OUT/pytest_trainer.log" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest_trainer.log"
timeout -k 10 200 python tools/diag_two_trainers.py > "$OUT/diag_two.json" 2> "$OUT/diag_two.err"; echo "diag rc=$?"; tail -c 1500 "$OUT/diag_two.json"
timeout -k 10 300 python bench.py --no-cpu-baseline --one-loop --steps 40 --warmup 10 > "$OUT/bench_eager.json" 2> "$OUT/bench_eager.err"; echo "bench eager rc=$?"
timeout -k 10 300 python bench.py --dist --no-cpu-baseline --one-loop --steps 40 --warmup 10 > "$OUT/bench_dist_eager.json" 2> "$OUT/bench_dist_eager.err"; echo "bench dist eager rc=$?"
timeout -k 10 300 python bench.py --dist --graph --no-cpu-baseline --one-loop --steps 40 --warmup 10 > "$OUT/bench_dist_graph.json" 2> "$OUT/bench_dist_graph.err"; echo "bench dist graph rc=$?"
timeout -k 10 300 python bench.py --dist --graph --amp bf16 --no-cpu-baseline --one-loop --steps 40 --warmup 10 > "$OUT/bench_dist_graph_bf16.json" 2> "$OUT/bench_dist_graph_bf16.err"; echo "bench dist graph bf16 rc=$?"
for f in bench_eager bench_dist_eager bench_dist_graph bench_dist_graph_bf16; do echo "== $f"; python - "$OUT/$f.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print({k:d.get(k) for k in ("value","ms_per_step","hip_graph","ranks_verified","gradient_exchange")})
    print("trainer_loop", d.get("trainer_loop"))
    print("roofline", {k:d.get("roofline",{}).get(k) for k in ("frac","launch_us","valu_issue_frac")})
except Exception as e:
    print("ERR", e); print(open(sys.argv[1].replace(".json",".err")).read()[-1500:])
PY
done
