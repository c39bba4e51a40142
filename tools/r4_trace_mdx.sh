#!/bin/bash
# per-call durations and grids of the hand-written kernels in one bench step (kernel trace, mdx:: rows only):
#   gpurun --timeout 900 -- 'bash tools/r4_trace_mdx.sh'  ->  gpurun_out/trace_mdx/mdx_calls.csv
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/trace_mdx"; rm -rf "$OUT"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$OUT/raw" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-trainer-loop --no-roofline --steps 6 --warmup 4 "$@" > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
f=$(find "$OUT/raw" -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$OUT/mdx_calls.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = csv.writer(open(sys.argv[2], "w"))
out.writerow(["kernel", "us", "grid", "wg"])
for r in rows:
    n = r["Kernel_Name"]
    if "mdx::" in n:
        out.writerow([n[:70], "%.2f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3),
                      "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]), "%sx%sx%s" % (r["Workgroup_Size_X"], r["Workgroup_Size_Y"], r["Workgroup_Size_Z"])])
PY
rm -rf "$OUT/raw"; wc -l "$OUT/mdx_calls.csv"
