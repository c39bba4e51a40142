#!/bin/bash
# Round-3 measurements that profiles/ and DESIGN.md quote.  On the GPU box:
#     gpurun --timeout 1200 -- 'bash tools/profile_round3.sh'
# then, back in the repo:      bash tools/profile_round3.sh --collect r03
# rocprofv3 rules on this pool: program directly after `--`, counters (--pmc) in their own passes without any trace
# option other than --kernel-trace, cwd /tmp.
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/profile_round3"
if [ "$1" = "--collect" ]; then
    tag="${2:-r03}"
    cd "$ROOT" || exit 1
    cp "$OUT/bench_kernel_pmc.json" "profiles/${tag}_bench_kernel_pmc.json"     # what bench.py quotes (tied to the library build by its hash)
    cp "$OUT/bench_full.json" "profiles/${tag}_bench_full.json"
    cp "$OUT/bench_bf16_dist_graph.json" "profiles/${tag}_bench_bf16_dist_graph.json"
    python tools/summarize_rocprof.py "$OUT/bench_stats" "profiles/${tag}_bench_kernel_stats.txt" \
        "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-trainer-loop --steps 30 --warmup 10   (MI355X, 1 GPU)" > /dev/null
    cp "$ROOT/gpurun_out/pmc_bench/summary.txt" "profiles/${tag}_bench_kernel_pmc.txt"
    cp "$OUT/kbench.txt" "profiles/${tag}_kbench.txt"
    ls -la profiles/ | grep "$tag"
    exit 0
fi
mkdir -p "$OUT" && cd /tmp && export TMPDIR=/tmp
# counters first: bench.py quotes them (for this very build of the library, taken on its own launches) next to the HBM fraction
bash "$ROOT/tools/pmc_bench.sh" > "$OUT/pmc_bench.log" 2>&1 || exit 1
python3 "$ROOT/tools/pmc_to_json.py" "$ROOT/gpurun_out/pmc_bench" "$ROOT/profiles/r03_bench_kernel_pmc.json" 12 192 640 2 4 > "$OUT/pmc_json.log" 2>&1 || exit 1
cp "$ROOT/profiles/r03_bench_kernel_pmc.json" "$OUT/bench_kernel_pmc.json"
python3 "$ROOT/tools/kbench.py" --what fwd,bwd,ident,train,pre --reps 20 > "$OUT/kbench.txt" 2>&1 || exit 1
python3 "$ROOT/bench.py" > "$OUT/bench_full.json" 2> "$OUT/bench_full.err" || exit 1
python3 "$ROOT/bench.py" --dist --graph --amp bf16 --no-cpu-baseline --one-loop > "$OUT/bench_bf16_dist_graph.json" 2> "$OUT/bench_bf16_dist_graph.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench_stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-trainer-loop --steps 30 --warmup 10 > "$OUT/bench_stats.log" 2>&1 || exit 1
tail -c 900 "$OUT/bench_full.json"
