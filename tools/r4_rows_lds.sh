#!/bin/bash
# horizontal pass, rows form: LDS budget per block (= columns per block, per scale) against the whole stage's time per batch
#   needs libmdx_ab_dev.so (MDX_BUILD_DEFINES=-DMDX_DEV_SWITCHES);  gpurun --timeout 900 -- 'bash tools/r4_rows_lds.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"; cd "$ROOT"
for rep in 1 2; do
for lds in 24000 32768 40960 45000 49152 57344 65536; do
    MDX_LIB="$PK/libmdx_ab_dev.so" MDX_RESAMPLE_ROWS_LDS=$lds timeout -k 10 200 python tools/loader_cost.py --samples 6 --reps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows LDS budget %6d   graph replay %.1f us per batch' % ($lds, d['gpu_us_per_batch']))"
done; done
