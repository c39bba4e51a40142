#!/bin/bash
# A/B of kernel builds: tools/kbench.py on every digging-..._amd/libmdx_ab_*.so (MDX_LIB selects the library)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
OUT="$ROOT/gpurun_out/ab"; mkdir -p "$OUT"; cd "$ROOT"
for lib in "$PK"/libmdx_hip.so "$PK"/libmdx_ab_*.so; do
    n=$(basename "$lib" .so)
    MDX_LIB="$lib" timeout -k 10 120 python tools/kbench.py --what train,pre --reps 30 > "$OUT/$n.txt" 2>&1
    echo "== $n"; grep -E "^train|^prologue" "$OUT/$n.txt"
done
