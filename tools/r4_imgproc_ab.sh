#!/bin/bash
# image preparation: the bit-exact tests, then gpu_us_per_batch (graph replay / eager) of every libmdx_hip.so / libmdx_ab_*.so, twice
#   gpurun --timeout 900 -- 'bash tools/r4_imgproc_ab.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"; cd "$ROOT"; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_imgproc.py -x -q -m gpu > gpurun_out/pytest_img.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 gpurun_out/pytest_img.log
[ $rc = 0 ] || exit 1
for rep in 1 2; do
    for lib in "$PK"/libmdx_hip.so "$PK"/libmdx_ab_*.so; do
        [ -f "$lib" ] || continue
        MDX_LIB="$lib" timeout -k 10 200 python tools/loader_cost.py --samples 6 --reps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s graph replay %.1f us  eager %.1f us per batch' % ('$(basename $lib .so)', d['gpu_us_per_batch'], d['gpu_us_per_batch_eager_loop']))"
    done
done
