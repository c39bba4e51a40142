#!/bin/bash
# The runs that settle what this build decided on ONE-rank evidence (no multi-GPU node was ever available to
# the builder, SCALE_r01..r03 are "skipped").  Needs an 8-GPU MI355X node; every line prints one JSON bench line into
# scale_check_out/.  Nothing here is run by the tests or by the driver -- it is the operator's checklist.
#
#   bash tools/scale_check.sh            # all sections;  SECTIONS="A C" bash tools/scale_check.sh   for some
#
# A. The contract's curve (what the driver's SCALE_rNN.json measures): eager data-parallel step, four 32 MB buckets issued
#    from inside backward, weak scaling (batch 12 per GPU).
#      expected: images/s(N) ~= N x images/s(1) x (0.95 .. 1.0).  The all-reduce of 107 MB (fp32) is link-bound at ~0.5-1 ms
#      on xGMI and overlaps backward; a step is ~20.5 ms (fp32) / ~13.3 ms (bf16 networks).
#      if efficiency at N = 8 is < 0.9: look at "gradient_exchange" in the line (buckets must be 4) and at the host --
#      an eager step enqueues ~1600 launches; with 8 ranks x (1 + workers) processes the host may be the limit, which is
#      what section B's captured step removes.
# B. The captured data-parallel step (opt-in: MDX_DP_GRAPH=1, which `bench.py --graph` sets for N > 1): the whole step,
#    RCCL exchange included, is ONE hipGraph per rank.  Never run with N > 1 so far (model_tool/parallel.py: dp_graph_allowed).
#      expected: >= the eager figure at every N (1-rank group: 576.7 vs 573.5 fp32, 894.8 vs ~850 bf16); ONE bucket per step,
#      not overlapped: costs ~0.5-1 ms per step at N = 8 (4-7 % of a bf16 step).  --grad-comm bf16 halves it.
#      a hang or a divergence here (loss of rank 0 differs from section A's by more than noise) means the capture of the
#      communicator's stream does not replay in the same order on every rank: keep the default (eager) and report.
#      if B beats A at N = 8 by > 3 %: make MDX_DP_GRAPH=1 the default in model_tool/parallel.py: dp_graph_allowed.
# C. Hardware queues for the captured step's process at N = 8 (GPU_MAX_HW_QUEUES; model_train.py asks for 2 when the step is
#    captured and WORLD_SIZE > 1 -- chosen with ONE rank's communicator, tools/sweep_hw_queues.sh).
#      expected reading: "trainer_loop.value" (the DataLoader-fed loop) highest at 2; if 4 (runtime default, MDX_HW_QUEUES=0)
#      or another value wins at N = 8, change the default in model_train.py / bench.py: trainer_loop_child.
# D. Can one host feed 8 ranks (no GPU needed): 8 DataLoader sets side by side.
#      expected: every set >= 1.3 x 900 samples/s (what a bf16 rank consumes); else the bottleneck is the host and
#      tools/prep_host_profile.py names the stage.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
OUT="$ROOT/scale_check_out"; mkdir -p "$OUT"
SECTIONS="${SECTIONS:-A B C D}"
run() { name=$1; shift; echo "== $name: $*"; "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "   rc=$? $(tail -c 400 "$OUT/$name.json" | tr '\n' ' ' | cut -c1-200)"; }
for sec in $SECTIONS; do case $sec in
A) for n in 1 2 4 8; do
       run "A_fp32_eager_n$n" python bench.py --gpus $n --steps 100 --warmup 20
       run "A_bf16_eager_n$n" python bench.py --gpus $n --steps 100 --warmup 20 --amp bf16
   done ;;
B) for n in 1 2 4 8; do
       run "B_fp32_graph_n$n" python bench.py --gpus $n --steps 100 --warmup 20 --graph
       run "B_bf16_graph_n$n" python bench.py --gpus $n --steps 100 --warmup 20 --graph --amp bf16
   done
   run "B_bf16_graph_n8_comm_bf16" python bench.py --gpus 8 --steps 100 --warmup 20 --graph --amp bf16 --grad-comm bf16 ;;
C) for q in 0 2 4; do
       MDX_HW_QUEUES=$q run "C_bf16_graph_n8_queues$q" python bench.py --gpus 8 --steps 60 --warmup 20 --graph --amp bf16 --one-loop
   done ;;
D) run "D_loader_8ranks" python tools/loader_cost.py --ranks 8 --workers 16 --seconds 15
   run "D_loader_6ranks_pinned" python tools/loader_cost.py --ranks 6 --workers 16 --seconds 15 --pin 1 ;;
esac; done
python - "$OUT" <<'PY'
import glob, json, os, sys
rows = []
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except (IndexError, ValueError):
        rows.append((os.path.basename(f), "no JSON line (see .err)")); continue
    if "value" in d:
        loop = d.get("trainer_loop", {})
        rows.append((os.path.basename(f), "%8.1f %s  n_gpus %s  graph %s  loop %s" % (d["value"], d.get("unit", ""), d.get("n_gpus"), d.get("hip_graph"), loop.get("value", loop.get("error", "-")))))
    else:
        rows.append((os.path.basename(f), "aggregate %s  min rank %s  meets need %s" % (d.get("aggregate_samples_per_s"), d.get("min_rank"), d.get("every_rank_meets_need"))))
for r in rows:
    print("%-34s %s" % r)
PY
