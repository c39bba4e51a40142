#!/usr/bin/env python3
"""Which convolution does each of MIOpen's zero-fills (SubTensorOpWithScalar1d in front of a split-K kernel) belong to?  From a
rocprofv3 kernel trace: for the last step, every zero-fill with the kernel that follows it on the same queue.

    python tools/zero_fills.py OUT_DIR [--min-us 8]
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
    ap.add_argument("--min-us", type=float, default=0.0)
    ap.add_argument("--marker", default="photometric_train_kernel")
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + a.dir)
    rows = []
    for r in csv.DictReader(open(max(files, key=os.path.getsize))):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    one = rows[marks[-2]:marks[-1]]
    per = collections.defaultdict(list)
    for r in one:
        per[r[3]].append(r)
    agg = collections.OrderedDict()
    total = 0.0
    for q, rs in per.items():
        for i, r in enumerate(rs[:-1]):
            if "SubTensorOpWithScalar" in r[2]:
                us = (r[1] - r[0]) / 1e3
                nxt = rs[i + 1]
                total += us
                key = nxt[2][:140]
                n, t, tn = agg.get(key, (0, 0.0, 0.0))
                agg[key] = (n + 1, t + us, tn + (nxt[1] - nxt[0]) / 1e3)
    print("zero-fills of the last step: %.1f us in all" % total)
    print("%5s %9s %9s  %s" % ("fills", "fill us", "kernel us", "the kernel behind them"))
    for k, (n, t, tn) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t >= a.min_us:
            print("%5d %9.1f %9.1f  %s" % (n, t, tn, k))
    return 0


if __name__ == "__main__":
    sys.exit(main())
