#!/bin/bash
# which stages should keep channels-last maps?  the step under several layout plans (mdx/layout.py), fp32 and bf16+graph
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/r5_plans"; mkdir -p "$OUT"; cd "$ROOT"
B="--no-cpu-baseline --no-trainer-loop --no-roofline --steps 30 --warmup 10"
for plan in "$@"; do
  tag=$(echo "$plan" | tr ',' '_')
  timeout -k 10 300 python bench.py $B --channels-last "$plan" > "$OUT/fp32_$tag.json" 2> "$OUT/fp32_$tag.err" || { tail -3 "$OUT/fp32_$tag.err"; exit 1; }
  timeout -k 10 300 python bench.py $B --channels-last "$plan" --amp bf16 --graph > "$OUT/bf16_$tag.json" 2> "$OUT/bf16_$tag.err" || { tail -3 "$OUT/bf16_$tag.err"; exit 1; }
  python - "$OUT" "$tag" <<'PY'
import json,sys,os
o,t=sys.argv[1:3]
r=[json.loads([l for l in open(os.path.join(o,"%s_%s.json"%(k,t))) if l.startswith("{")][-1]) for k in ("fp32","bf16")]
print("%-40s fp32 %7.1f img/s %.3f ms | bf16+graph %7.1f img/s %.3f ms" % (t, r[0]["value"], r[0]["ms_per_step"], r[1]["value"], r[1]["ms_per_step"]))
PY
done
