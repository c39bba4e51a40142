#!/bin/bash
# Record MIOpen find-db entries for one configuration of the step and measure with them:
#   gpurun --timeout 1200 -- 'bash tools/find_db.sh <tag> [bench.py args, e.g. --channels-last --amp bf16 --graph]'
# The shipped db (…_amd/miopen_db) is copied to gpurun_out/miopen_db_<tag>/ and MIOpen appends what it finds there; merge
# the new lines back with tools/merge_find_db.py.  A search of ~70 convolutions x 3 directions takes 10-15 minutes.
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
TAG="${1:-nhwc}"; shift
PKG="$ROOT/digging-into-self-supervised-monocular-depth-estimation_amd"
DB="$ROOT/gpurun_out/miopen_db_$TAG"; OUT="$ROOT/gpurun_out/find_$TAG"; mkdir -p "$DB" "$OUT"; cd "$ROOT"
[ -n "$(ls "$DB" 2>/dev/null)" ] || { cp "$PKG"/miopen_db/*.txt "$DB"/; mkdir -p "$DB/base"; cp "$PKG"/miopen_db/*.txt "$DB/base"/; }
export MIOPEN_USER_DB_PATH="$DB"
B="--no-cpu-baseline --no-trainer-loop --no-roofline"
( while sleep 45; do echo "[find] $(date +%T) db lines: $(cat "$DB"/*.ufdb.txt | wc -l)"; done ) & TICK=$!
timeout -k 10 "${FIND_LIMIT:-900}" python bench.py $B --steps 3 --warmup 2 --miopen-find "$@" > "$OUT/find.json" 2> "$OUT/find.err"; rc=$?
kill $TICK
echo "find rc=$rc; db lines: $(cat "$DB"/*.ufdb.txt | wc -l)"
[ $rc -ne 0 ] && { tail -5 "$OUT/find.err"; exit $rc; }
timeout -k 10 200 python bench.py $B --steps 30 --warmup 10 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
grep '^{' "$OUT/bench.json" | tail -1 | cut -c1-160
