#!/bin/bash
# rows form of the horizontal pass: LDS budget per block (= columns per block) against time, per scale of the KITTI pyramid
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
export MDX_LIB="$PK/libmdx_ab_dev.so"
cd "$ROOT"
for cfg in "s0 32 192x640" "s1 12 96x320" "s2 12 48x160" "s3 12 24x80"; do
    set -- $cfg
    for lds in 24000 32768 40960 45000 49152 65536; do
        MDX_RESAMPLE_ROWS_LDS=$lds bash tools/trace_imgbench.sh sw_$1 --n $2 --out $3 | grep resample_h | sed "s/^/lds $lds  /"
    done
done
