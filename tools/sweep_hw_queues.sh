#!/bin/bash
# trainer loop (tools/loop_bisect.py) against the number of hardware queues the HIP runtime may use (GPU_MAX_HW_QUEUES, default 4)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
for amp in none bf16; do for dist in 0 1; do for q in ${QUEUES:-2 4}; do
    echo "== amp=$amp dist=$dist GPU_MAX_HW_QUEUES=$q"
    GPU_MAX_HW_QUEUES=$q DIST=$dist AMP=$amp RAW=1 timeout -k 10 400 python tools/loop_bisect.py 2>&1 | grep "ms/step" | sed -n '1p;4p;5p'
done; done; done
