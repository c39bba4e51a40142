#!/bin/bash
# Refreshes the measurements that profiles/ and DESIGN.md quote, on the GPU box:
#     gpurun --timeout 1100 -- 'bash tools/profile_round.sh'
# then, back in the repo:   bash tools/profile_round.sh --collect r01
# rocprofv3 rules on this pool: program directly after `--`, counters (--pmc) in their own passes without any
# trace option other than --kernel-trace, cwd /tmp.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/profile_round"
if [ "$1" = "--collect" ]; then
    tag="${2:-r01}"
    cd "$ROOT" || exit 1
    cp "$OUT/bench_unprofiled.json" "profiles/${tag}_bench_unprofiled.json"
    python tools/summarize_rocprof.py "$OUT/bench_stats" "profiles/${tag}_bench_kernel_stats.txt" \
        "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline   (MI355X, 1 GPU)" > /dev/null
    python tools/summarize_rocprof.py "$OUT/kbench_stats" "profiles/${tag}_kbench_kernel_stats.txt" \
        "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/kbench.py   (hand-written kernels only; B=12 192x640 S=2)" > /dev/null
    python tools/pmc_summary.py "profiles/${tag}_kernel_pmc.txt" "$OUT"/pmc_* > /dev/null
    ls -la profiles/
    exit 0
fi
mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
if [ "$1" = "--bench-stats-only" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1
exit $?
fi
if [ "$1" != "--pmc-only" ]; then
python3 "$ROOT/bench.py" > "$OUT/bench_full.json" 2> "$OUT/bench_full.err" || exit 1
python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline > "$OUT/bench_stats.log" 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kbench_stats" -- python3 "$ROOT/tools/kbench.py" > "$OUT/kbench_stats.log" 2>&1 || exit 1
fi
# FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950 ("exceeds the capabilities of the hardware"): one each
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum" "TCC_MISS_sum"; do
    i=$((i + 1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/pmc_$i" -- python3 "$ROOT/tools/kbench.py" --what fwd,bwd,ident --reps 6 > "$OUT/pmc_$i.log" 2>&1
    echo "pmc pass $i ($grp): rc=$?"
done
tail -c 400 "$OUT/bench_full.json"
