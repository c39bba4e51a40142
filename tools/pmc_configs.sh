#!/bin/bash
# Round 4: counters of the training kernel on BASELINE configs[1] (the default bench), configs[4]
# (<3,true,true>, mono+stereo) and configs[3] (<2,true,true> at 8x320x1024) -- the timed step's own launches (tools/pmc_bench.sh).
#   then: python tools/pmc_to_json.py gpurun_out/pmc_bench profiles/r04_bench_kernel_pmc.json 12 192 640 2 4
#         python tools/pmc_to_json.py gpurun_out/pmc_c4 profiles/r04_pmc_c4.json 12 192 640 3 4
#         python tools/pmc_to_json.py gpurun_out/pmc_c3 profiles/r04_pmc_c3.json 8 320 1024 2 4
#   gpurun --timeout 1150 -- 'bash tools/r4_pmc_configs.sh'
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$ROOT/gpurun_out"
PMC_OUT=pmc_bench PMC_SHAPE="B=12, 192x640, S=2" bash "$ROOT/tools/pmc_bench.sh" > "$ROOT/gpurun_out/pmc_bench.log" 2>&1 || exit 1
tail -40 "$ROOT/gpurun_out/pmc_bench.log"
PMC_OUT=pmc_c4 PMC_SHAPE="B=12, 192x640, S=3" bash "$ROOT/tools/pmc_bench.sh" --frame-ids "0 -1 1 s" > "$ROOT/gpurun_out/pmc_c4.log" 2>&1 || exit 1
tail -60 "$ROOT/gpurun_out/pmc_c4.log"
PMC_OUT=pmc_c3 PMC_SHAPE="B=8, 320x1024, S=2" bash "$ROOT/tools/pmc_bench.sh" --height 320 --width 1024 --num-layers 50 --batch 8 --amp bf16 > "$ROOT/gpurun_out/pmc_c3.log" 2>&1 || exit 1
tail -60 "$ROOT/gpurun_out/pmc_c3.log"
