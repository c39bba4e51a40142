#!/bin/bash
# counters of the training kernel's two forms (ident + noise / fed by the prologue) on tools/kbench.py's data
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_pre"; rm -rf "$OUT"; mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_WAIT_INST_LDS"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --kernel-include-regex "photometric_train_kernel" --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/tools/kbench.py" --what train,pre --reps 4 > "$OUT/pmc_$i.log" 2>&1
    echo "pmc pass $i ($grp): rc=$?"
done
MDX_PMC_TITLE="tools/kbench.py --what train,pre" python3 "$ROOT/tools/pmc_summary.py" "$OUT/summary.txt" "$OUT"/pmc_* > /dev/null
cat "$OUT/summary.txt"
