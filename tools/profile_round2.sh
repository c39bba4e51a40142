#!/bin/bash
# Round-2 measurements that profiles/ and DESIGN.md quote.  On the GPU box:
#     gpurun --timeout 1100 -- 'bash tools/profile_round2.sh'
# then, back in the repo:      bash tools/profile_round2.sh --collect r02
# rocprofv3 rules on this pool: program directly after `--`, counters (--pmc) in their own passes without any trace
# option other than --kernel-trace, cwd /tmp.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/profile_round2"
if [ "$1" = "--collect" ]; then
    tag="${2:-r02}"
    cd "$ROOT" || exit 1
    cp "$OUT/bench_full.json" "profiles/${tag}_bench_full.json"
    cp "$OUT/bench_per_scale.json" "profiles/${tag}_bench_per_scale_kernels.json"
    python tools/summarize_rocprof.py "$OUT/bench_stats" "profiles/${tag}_bench_kernel_stats.txt" \
        "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-trainer-loop   (MI355X, 1 GPU)" > /dev/null
    python tools/summarize_rocprof.py "$ROOT/gpurun_out/pmc_train/trace" "profiles/${tag}_kbench_train_kernel_stats.txt" \
        "rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/kbench.py --what train   (B=12 192x640 S=2, 4 scales)" > /dev/null
    cp "$ROOT/gpurun_out/pmc_train/summary.txt" "profiles/${tag}_train_kernel_pmc.txt"
    cp "$OUT/train_kernel_pmc.json" "profiles/${tag}_train_kernel_pmc.json"
    cp "$OUT/kbench.txt" "profiles/${tag}_kbench.txt"
    ls -la profiles/
    exit 0
fi
mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
# counters first: bench.py quotes them (for this very build of the library) next to the HBM fraction
python3 "$ROOT/tools/kbench.py" --what fwd,bwd,ident,train --reps 20 > "$OUT/kbench.txt" 2>&1 || exit 1
bash "$ROOT/tools/pmc_train.sh" > "$OUT/pmc_train.log" 2>&1 || exit 1
python3 "$ROOT/tools/pmc_to_json.py" "$ROOT/gpurun_out/pmc_train" "$OUT/train_kernel_pmc.json" 12 192 640 2 4 > "$OUT/pmc_json.log" 2>&1 || exit 1
cp "$OUT/train_kernel_pmc.json" "$ROOT/profiles/r02_train_kernel_pmc.json"
python3 "$ROOT/bench.py" > "$OUT/bench_full.json" 2> "$OUT/bench_full.err" || exit 1
python3 "$ROOT/bench.py" --no-cpu-baseline --no-trainer-loop --per-scale-kernels > "$OUT/bench_per_scale.json" 2> "$OUT/bench_per_scale.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-trainer-loop > "$OUT/bench_stats.log" 2>&1 || exit 1
tail -c 600 "$OUT/bench_full.json"
