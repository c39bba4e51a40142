#!/bin/bash
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r3c"; mkdir -p "$OUT"; cd "$ROOT"
timeout -k 10 900 python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?"; tail -4 "$OUT/pytest_gpu.log"
timeout -k 10 300 python tools/diag_grad_elementwise.py > "$OUT/grad_elementwise.txt" 2> "$OUT/grad_elementwise.err"; echo "diag rc=$?"
B="python bench.py --no-cpu-baseline --no-trainer-loop --no-roofline --steps 40 --warmup 10"
run() { name=$1; shift; timeout -k 10 200 env "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$? $(python -c "import json,sys; d=json.loads([l for l in open('$OUT/$name.json') if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d.get('gradient_exchange',{}) and d['gradient_exchange'].get('buckets'))" 2>&1 | tail -1)"; }
run graph_plain X=1 $B --graph
run dist_graph_nocomm MDX_SYNC_NO_COMM=1 $B --dist --graph
run dist_graph X=1 $B --dist --graph
run dist_eager X=1 $B --dist
run dist_eager_nocomm MDX_SYNC_NO_COMM=1 $B --dist
run bf16_graph_plain X=1 $B --graph --amp bf16
run bf16_dist_graph X=1 $B --dist --graph --amp bf16
run bf16_dist_graph_bf16comm X=1 $B --dist --graph --amp bf16 --grad-comm bf16
