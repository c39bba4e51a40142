#!/bin/bash
# The runs that settle what this build decided on ONE-rank evidence (no multi-GPU node was ever available to
# the builder, SCALE_r01..r03 are "skipped").  Needs an 8-GPU MI355X node; every line prints one JSON bench line into
# scale_check_out/.  Nothing here is run by the tests or by the driver -- it is the operator's checklist.
#
#   bash tools/scale_check.sh            # all sections;  SECTIONS="A C" bash tools/scale_check.sh   for some
#
# A. The contract's curve (what the driver's SCALE_rNN.json measures: `python bench.py --gpus N`).  Since round 5 that is the
#    SPLIT captured step: graph A = forward + backward (pose network beside the depth network on a side stream) + gradients
#    gathered into the flat buffer; ONE all-reduce issued eagerly between the replays (RCCL is never inside a capture); graph B =
#    Adam.  Weak scaling, batch 12 per GPU.
#      expected: images/s(N) ~= N x images/s(1) x (0.92 .. 0.98): the all-reduce of 107 MB (fp32) is link-bound at ~0.6-1.3 ms on
#      xGMI and NOT overlapped with backward -- 4-8 % of a 16 ms fp32 step, 8-15 % of an 8 ms bf16 step (--grad-comm bf16 halves
#      it).  "hip_graph_form" in the line must say "split", "rank_ms_per_step" shows a straggler.
#      if efficiency at N = 8 is far below that: compare with section E (eager, four buckets overlapping backward).
# B. The whole step INCLUDING the exchange as ONE hipGraph per rank (MDX_DP_GRAPH=1).  Never run with N > 1 so far
#    (model_tool/parallel.py: dp_graph_allowed); with one rank it is 0.9 % ahead of the split form (805.8 vs 799.0 images/s).
#      a hang or a divergence here (rank 0's loss differs from section A's by more than noise) means the capture of the
#      communicator's stream does not replay in the same order on every rank: keep the split form and report.
#      if B beats A at N = 8 by > 3 %: make MDX_DP_GRAPH=1 the default in model_tool/parallel.py: dp_graph_allowed.
# C. Hardware queues for the captured step's process at N = 8 (GPU_MAX_HW_QUEUES; bench.py / model_train.py ask for 2 -- chosen
#    with ONE rank's communicator: resident | loop 742 | 722 images/s at 2, 744 | 669 at 4, 514 | 723 at 8).
#      expected reading: "value" and "trainer_loop.value" highest at 2; if 4 (runtime default, MDX_HW_QUEUES=0) or 8 wins at
#      N = 8, change the default in model_train.py / bench.py.
# D. Can one host feed 8 ranks (no GPU needed): 8 DataLoader sets side by side; prints PASS / FAIL against 1.3 x what a bf16
#    rank consumes (~1780 samples/s since round 5; pass --consume to change).
# E. The eager data-parallel step (four 32 MB buckets issued from inside backward, overlapping it): host-bound since round 5
#    (~900 launches per 15 ms step; 421-810 images/s on one GPU depending on the host), kept as the fallback.
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
OUT="$ROOT/scale_check_out"; mkdir -p "$OUT"
SECTIONS="${SECTIONS:-A B C D E}"
run() { name=$1; shift; echo "== $name: $*"; "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "   rc=$? $(tail -c 400 "$OUT/$name.json" | tr '\n' ' ' | cut -c1-200)"; }
for sec in $SECTIONS; do case $sec in
A) for n in 1 2 4 8; do
       run "A_fp32_split_n$n" python bench.py --gpus $n --steps 100 --warmup 20
       run "A_bf16_split_n$n" python bench.py --gpus $n --steps 100 --warmup 20 --amp bf16
   done
   run "A_bf16_split_n8_comm_bf16" python bench.py --gpus 8 --steps 100 --warmup 20 --amp bf16 --grad-comm bf16 ;;
B) for n in 2 4 8; do
       MDX_DP_GRAPH=1 run "B_fp32_onegraph_n$n" python bench.py --gpus $n --steps 100 --warmup 20
       MDX_DP_GRAPH=1 run "B_bf16_onegraph_n$n" python bench.py --gpus $n --steps 100 --warmup 20 --amp bf16
   done ;;
C) for q in 0 2 8; do
       MDX_HW_QUEUES=$q run "C_bf16_split_n8_queues$q" python bench.py --gpus 8 --steps 60 --warmup 20 --amp bf16 --one-loop
   done ;;
D) run "D_loader_8ranks" python tools/loader_cost.py --ranks 8 --workers 24 --seconds 15
   run "D_loader_6ranks_pinned" python tools/loader_cost.py --ranks 6 --workers 24 --seconds 15 --pin 1 ;;
E) for n in 1 8; do
       run "E_fp32_eager_n$n" python bench.py --gpus $n --steps 100 --warmup 20 --eager
       MDX_HW_QUEUES=8 run "E_fp32_eager_n${n}_queues8" python bench.py --gpus $n --steps 100 --warmup 20 --eager
   done ;;
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
