#!/usr/bin/env python3
"""Where the WALL time of a replayed step goes, from a rocprofv3 kernel trace: how long 0, 1, 2, ... kernels are in flight, and
what each hardware queue (one per branch of the captured step: depth network / pose network) does -- its first start, last end,
busy time and launch count.  Steps are delimited by the training kernel (one launch per step).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --no-cpu-baseline --no-trainer-loop --no-roofline --steps 12 --warmup 4
    python tools/timeline.py OUT [--steps 8]
"""
import argparse
import collections
import csv
import glob
import os
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--steps", type=int, default=8, help="the last N whole steps of the trace")
    ap.add_argument("--marker", default="photometric_train_kernel")
    ap.add_argument("--gaps", type=int, default=12, help="print the kernels in front of the N longest idle gaps")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + a.dir)
    rows = []
    for r in csv.DictReader(open(max(files, key=os.path.getsize))):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < a.steps + 2:
        sys.exit("only %d steps in the trace" % len(marks))
    # a step = from one marker's start to the next one's: the marker sits at a fixed place of every step
    lo, hi = marks[-a.steps - 1], marks[-1]
    span = rows[lo:hi]
    t0, t1 = span[0][0], rows[hi][0]
    wall = (t1 - t0) / a.steps
    # in-flight histogram
    ev = []
    for s, e, *_ in span:
        ev.append((s, 1))
        ev.append((min(e, t1), -1))
    ev.sort()
    depth, last, hist = 0, t0, collections.Counter()
    for t, d in ev:
        hist[depth] += t - last
        depth, last = depth + d, t
    print("steps %d   wall %.3f ms/step   kernels %.1f/step   kernel time %.3f ms/step" %
          (a.steps, wall / 1e6, len(span) / a.steps, sum(min(e, t1) - s for s, e, *_ in span) / a.steps / 1e6))
    print("kernels in flight : share of the wall time")
    for k in sorted(hist):
        print("   %d : %6.3f ms/step  %5.1f %%" % (k, hist[k] / a.steps / 1e6, 100.0 * hist[k] / (t1 - t0)))
    # per queue
    per = collections.defaultdict(list)
    for r in span:
        per[(r[3], r[4])].append(r)
    print("queue/stream      launches/step  busy ms/step   mean gap us   (gap = idle time between two consecutive kernels of the queue)")
    for q, rs in sorted(per.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, *_ in rs)
        gaps = [max(0, rs[i + 1][0] - rs[i][1]) for i in range(len(rs) - 1)]
        small = [g for g in gaps if g < 50000]
        print("   %-14s %8.1f %12.3f %12.2f" % ("%s/%s" % q, len(rs) / a.steps, busy / a.steps / 1e6, (sum(small) / max(1, len(small))) / 1e3))
    # the last step alone: who ends last, and the longest idle gaps
    one = rows[marks[-2]:marks[-1]]
    ends = collections.defaultdict(int)
    for s, e, n, q, st in one:
        ends[(q, st)] = max(ends[(q, st)], e)
    base = one[0][0]
    print("last step: queue -> last kernel end (ms after the training kernel's start)")
    for q, e in sorted(ends.items(), key=lambda kv: kv[1]):
        print("   %-14s %8.3f" % ("%s/%s" % q, (e - base) / 1e6))
    idle, cur_end = [], one[0][1]
    for i in range(1, len(one)):
        s, e, n, q, st = one[i]
        if s > cur_end:
            idle.append((s - cur_end, i))
        cur_end = max(cur_end, e)
    idle.sort(reverse=True)
    print("last step: %d idle gaps, %.3f ms in all; the longest:" % (len(idle), sum(g for g, _ in idle) / 1e6))
    for g, i in idle[:a.gaps]:
        print("   %6.2f us  after %-60s before %s" % (g / 1e3, one[i - 1][2][:60], one[i][2][:60]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
