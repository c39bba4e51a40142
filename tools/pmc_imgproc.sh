#!/bin/bash
# PMC passes over tools/imgbench.py (the image-preparation kernels alone).
#   gpurun --timeout 900 -- 'bash tools/pmc_imgproc.sh [imgbench args]'
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/pmc_imgproc"
rm -rf "$OUT"; mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/tools/imgbench.py" "$@" > "$OUT/imgbench.txt" 2>&1 || exit 1
cat "$OUT/imgbench.txt"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/tools/imgbench.py" --reps 3 "$@" > "$OUT/pmc_$i.log" 2>&1
    echo "pmc pass $i ($grp): rc=$?"
done
MDX_PMC_FILTER="mdx::resample,mdx::jitter" python3 "$ROOT/tools/pmc_summary.py" "$OUT/summary.txt" "$OUT"/pmc_* > /dev/null
cat "$OUT/summary.txt"
