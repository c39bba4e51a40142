#!/bin/bash
# where does the captured data-parallel step lose time?  (1-rank RCCL group)
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r3b"; mkdir -p "$OUT"; cd "$ROOT"
B="python bench.py --no-cpu-baseline --no-trainer-loop --no-roofline --steps 40 --warmup 10"
run() { name=$1; shift; timeout -k 10 200 env "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$? $(python -c "import json,sys; d=json.loads([l for l in open('$OUT/$name.json') if l.startswith('{')][-1]); print(round(d['value'],1), round(d['ms_per_step'],3))" 2>&1 | tail -1)"; }
run graph_plain X=1 $B --graph
run dist_graph_nocomm MDX_SYNC_NO_COMM=1 $B --dist --graph
run dist_graph_1bucket X=1 $B --dist --graph --bucket-mb 4096
run dist_graph_blocking MDX_SYNC_BLOCKING=1 $B --dist --graph
run dist_graph_4buckets X=1 $B --dist --graph
run dist_eager_nocomm MDX_SYNC_NO_COMM=1 $B --dist
run bf16_graph_plain X=1 $B --graph --amp bf16
run bf16_dist_graph_nocomm MDX_SYNC_NO_COMM=1 $B --dist --graph --amp bf16
run bf16_dist_graph_1bucket X=1 $B --dist --graph --amp bf16 --bucket-mb 4096
