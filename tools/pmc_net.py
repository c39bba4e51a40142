#!/usr/bin/env python3
"""Counters of the hand-written NETWORK kernels (batch norm + add + ReLU, decoder glue, max-pool) per kernel AND launch geometry --
their maps differ from layer to layer, a per-kernel mean would say nothing -- from the passes of `PMC_NET=1 tools/pmc_bench.sh`:

    python tools/pmc_net.py gpurun_out/pmc_net profiles/r05_net_kernel_pmc.txt "title" [profiles/r05_net_kernel_pmc.json]

Per (kernel, grid): launches per step, mean duration (the kernel-trace pass), HBM-side traffic per launch as MI355X_MICROARCH.md
prescribes (FETCH_SIZE and WRITE_SIZE from separate passes, KiB; FETCH_SIZE x 2 on gfx950 -- calibrated for wide coalesced
reads, which is what these kernels issue: 16 bytes per lane), traffic / time, and its fraction of the 8 TB/s peak.  These are
streaming kernels: the HBM roof is their roof.  (Maps that fit the 256 MB Infinity Cache are partly served from it: the counters
then show LESS than the bytes the kernel touched, and a rate above what HBM alone delivers is not an error.)"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    src, dst, title = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if name.startswith("mdx::"):
                cnt[(name, r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    steps = 1
    for f in glob.glob(os.path.join(src, "trace", "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if name.startswith("mdx::"):
                grid = str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) if "Grid_Size_X" in r else r.get("Grid_Size", "")
                dur[(name, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            if "photometric_train_kernel" in name:
                steps += 1
    steps = max(1, steps - 1)
    rows = []
    for key, d in cnt.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        if key not in dur or "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
            continue
        us = sum(dur[key]) / len(dur[key])
        traffic = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        rows.append({"kernel": key[0], "grid": key[1], "launches_per_step": len(dur[key]) / steps, "us": us,
                     "fetch_mb": 2 * m["FETCH_SIZE"] * 1024 / 1e6, "write_mb": m["WRITE_SIZE"] * 1024 / 1e6,
                     "tb_s": traffic / us / 1e6, "frac_of_8tb_s": traffic / us / 1e6 / 8.0,
                     "ms_per_step": us * len(dur[key]) / steps / 1e3})
    rows.sort(key=lambda r: -r["ms_per_step"])
    lines = ["# " + title, "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) + kernel trace; per (kernel, grid size); traffic = 2 x FETCH + WRITE",
             "%-58s %10s %7s %8s %9s %9s %7s %6s %8s" % ("kernel", "grid", "n/step", "us", "fetch MB", "write MB", "TB/s", "frac", "ms/step")]
    fam = collections.Counter()
    for r in rows:
        lines.append("%-58s %10s %7.1f %8.1f %9.1f %9.1f %7.2f %6.2f %8.3f" % (r["kernel"].replace("mdx::", "")[:58], r["grid"], r["launches_per_step"],
                     r["us"], r["fetch_mb"], r["write_mb"], r["tb_s"], r["frac_of_8tb_s"], r["ms_per_step"]))
        fam["bn" if "bn_" in r["kernel"] else "glue" if ("glue" in r["kernel"] or "colsum" in r["kernel"]) else "maxpool"] += r["ms_per_step"]
    lines.append("# per step: " + ", ".join("%s %.3f ms" % kv for kv in fam.items()))
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:60]))
    if len(sys.argv) > 4:
        json.dump({"title": title, "rows": rows, "family_ms_per_step": dict(fam)}, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
