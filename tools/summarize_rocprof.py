#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats (csv) output directory into a small text summary that can be
committed under profiles/: every hand-written (mdx::) kernel plus the top N others.

    python tools/summarize_rocprof.py gpurun_out/prof2 profiles/r01_bench_kernel_stats.txt "command line"
"""
import csv
import glob
import os
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    # gpurun merges every call's files into the same local directory: take the newest run
    f = max(glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary", "# command: " + cmd,
             "# total kernel time %.3f ms over %d distinct kernels" % (tot / 1e6, len(rows)),
             "%-92s %7s %11s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%")]
    mdx = [r for r in rows if "mdx::" in r["Name"]]
    others = [r for r in rows if "mdx::" not in r["Name"]][:60]
    for group, title in ((mdx, "## hand-written kernels (libmdx_hip.so)"), (others, "## top 60 other kernels (MIOpen / ATen)")):
        lines.append(title)
        for r in group:
            lines.append("%-92s %7s %11.3f %10.2f %10.2f %10.2f %6.2f" % (
                r["Name"][:92], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    main()
