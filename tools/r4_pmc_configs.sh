#!/bin/bash
# Round 4: counters of the training kernel on BASELINE configs[4] (<3,true,true>, mono+stereo) and
# configs[3] (<2,true,true> at 8x320x1024) -- the timed step's own launches, as tools/pmc_bench.sh does for configs[1].
#   gpurun --timeout 1150 -- 'bash tools/r4_pmc_configs.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
PMC_OUT=pmc_c4 PMC_SHAPE="B=12, 192x640, S=3" bash "$ROOT/tools/pmc_bench.sh" --frame-ids "0 -1 1 s" > "$ROOT/gpurun_out/pmc_c4.log" 2>&1 || exit 1
tail -60 "$ROOT/gpurun_out/pmc_c4.log"
PMC_OUT=pmc_c3 PMC_SHAPE="B=8, 320x1024, S=2" bash "$ROOT/tools/pmc_bench.sh" --height 320 --width 1024 --num-layers 50 --batch 8 --amp bf16 > "$ROOT/gpurun_out/pmc_c3.log" 2>&1 || exit 1
tail -60 "$ROOT/gpurun_out/pmc_c3.log"
