#!/bin/bash
# per-kernel durations of tools/imgbench.py from a rocprofv3 kernel trace:  bash tools/trace_imgbench.sh <tag> [imgbench args]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
tag=$1; shift
OUT="$ROOT/gpurun_out/trace_img_$tag"
rm -rf "$OUT"; mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$ROOT/tools/imgbench.py" --reps 10 "$@" > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" "$tag" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mdx::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    v = v[3:]
    print("%-10s %-50s n=%3d  avg %7.1f us  min %7.1f" % (sys.argv[2], k, len(v), sum(v) / len(v), min(v)))
PY
