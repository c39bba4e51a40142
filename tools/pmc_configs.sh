#!/bin/bash
# Counter passes over the bench's own launches (tools/pmc_bench.sh) for the three BASELINE workload shapes of the training kernel
# -- configs[1] (the default bench), configs[4] (mono + stereo, <3,true,true>) and configs[3] (8x320x1024, ResNet-50, bf16) -- and for
# the network kernels between the convolutions at configs[1] fp32 and configs[3] bf16.
#   gpurun --timeout 1200 -- 'bash tools/pmc_configs.sh [train|net|all]'     then, here:
#   python tools/pmc_to_json.py gpurun_out/pmc_bench profiles/r05_bench_kernel_pmc.json 12 192 640 2 4
#   python tools/pmc_to_json.py gpurun_out/pmc_c4 profiles/r05_pmc_c4.json 12 192 640 3 4
#   python tools/pmc_to_json.py gpurun_out/pmc_c3 profiles/r05_pmc_c3.json 8 320 1024 2 4
#   cp gpurun_out/pmc_*/summary.txt -> profiles/r05_*_pmc.txt ; gpurun_out/pmc_net*/summary.txt -> profiles/r05_net_kernel_pmc*.txt
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
WHAT="${1:-all}"
mkdir -p "$ROOT/gpurun_out"
if [ "$WHAT" = train ] || [ "$WHAT" = all ]; then
PMC_OUT=pmc_bench PMC_SHAPE="B=12, 192x640, S=2" bash "$ROOT/tools/pmc_bench.sh" > "$ROOT/gpurun_out/pmc_bench.log" 2>&1 || exit 1
tail -30 "$ROOT/gpurun_out/pmc_bench.log"
PMC_OUT=pmc_c4 PMC_SHAPE="B=12, 192x640, S=3" bash "$ROOT/tools/pmc_bench.sh" --frame-ids "0 -1 1 s" > "$ROOT/gpurun_out/pmc_c4.log" 2>&1 || exit 1
tail -30 "$ROOT/gpurun_out/pmc_c4.log"
PMC_OUT=pmc_c3 PMC_SHAPE="B=8, 320x1024, S=2" bash "$ROOT/tools/pmc_bench.sh" --height 320 --width 1024 --num-layers 50 --batch 8 --amp bf16 > "$ROOT/gpurun_out/pmc_c3.log" 2>&1 || exit 1
tail -30 "$ROOT/gpurun_out/pmc_c3.log"
fi
if [ "$WHAT" = net ] || [ "$WHAT" = all ]; then
PMC_NET=1 PMC_OUT=pmc_net PMC_SHAPE="configs[1]: B=12, 192x640, ResNet-18, float32, channels-last" bash "$ROOT/tools/pmc_bench.sh" > "$ROOT/gpurun_out/pmc_net.log" 2>&1 || exit 1
tail -45 "$ROOT/gpurun_out/pmc_net.log"
PMC_NET=1 PMC_OUT=pmc_net_c3 PMC_SHAPE="configs[3]: B=8, 320x1024, ResNet-50, bfloat16, channels-last" bash "$ROOT/tools/pmc_bench.sh" --height 320 --width 1024 --num-layers 50 --batch 8 --amp bf16 > "$ROOT/gpurun_out/pmc_net_c3.log" 2>&1 || exit 1
tail -45 "$ROOT/gpurun_out/pmc_net_c3.log"
fi
