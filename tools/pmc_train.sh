#!/bin/bash
# PMC + kernel-trace passes over tools/kbench.py --what train (the one-launch training kernel alone).
#   gpurun --timeout 900 -- 'bash tools/pmc_train.sh [extra kbench args]'
# rocprofv3 rules on this pool: program directly after `--`, counters in their own passes, cwd /tmp.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_train"
rm -rf "$OUT"; mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/kbench.py" --what train --reps 10 "$@" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/tools/kbench.py" --what train --reps 4 "$@" > "$OUT/pmc_$i.log" 2>&1
    echo "pmc pass $i ($grp): rc=$?"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT/summary.txt" "$OUT"/pmc_* > /dev/null
cat "$OUT/summary.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("%-72s n=%4d avg %8.1f us  total %9.1f us" % (k, len(v), sum(v) / len(v), sum(v)))
PY
