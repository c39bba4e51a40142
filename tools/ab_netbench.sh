#!/bin/bash
# A/B of library builds on the network kernels' per-shape times: tools/netbench.py under every digging-..._amd/libmdx_ab_*.so
# (MDX_LIB selects the library) and the shipped one.   gpurun -- 'bash tools/ab_netbench.sh [netbench args, e.g. --only bn --bf16 --config 3]'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PK="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"; OUT="$ROOT/gpurun_out/ab_net"; mkdir -p "$OUT"; cd "$ROOT"
for lib in "$PK"/libmdx_hip.so "$PK"/libmdx_ab_*.so; do
  n=$(basename "$lib" .so)
  MDX_LIB="$lib" timeout -k 10 300 python tools/netbench.py "$@" > "$OUT/$n.txt" 2>&1 || { echo "$n FAILED"; tail -3 "$OUT/$n.txt"; continue; }
  printf "%-16s %s\n" "$n" "$(grep '^sum over' "$OUT/$n.txt")"
done
