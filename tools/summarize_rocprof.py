#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats (csv) output directory into a small text summary that can be
committed under profiles/: every hand-written (mdx::) kernel plus the top N others.

    python tools/summarize_rocprof.py gpurun_out/prof2 profiles/r01_bench_kernel_stats.txt "command line"
"""
import csv
import glob
import os
import sys


FAMILIES = (("miopen conv:fwd", ("igemm_fwd", "conv_fwd", "ConvFwd")), ("miopen conv:bwd", ("igemm_bwd", "conv_bwd_data", "ConvBwd")),
            ("miopen conv:wrw", ("igemm_wrw", "conv_bwd_weight", "ConvWrw")), ("miopen zero-fill", ("SubTensorOpWithScalar",)),
            ("mdx:bn_", ("bn_nhwc", "bn_act", "mdx::bn_")), ("mdx:decoder_glue", ("decoder_glue",)), ("mdx:maxpool", ("maxpool3s2",)),
            ("mdx:disp_head", ("disp_head",)), ("mdx:pose head / input", ("bias_act", "mean_bias", "colsum_scale", "encoder_input",
                                                                      "pose_projection", "param2matrix", "compose_projection")),
            ("mdx:loss path", ("photometric", "smooth_multi", "train_finish", "loss_total")), ("mdx:adam", ("adam_step",)), ("mdx:thin conv wgrad (MFMA)", ("thin_wgrad",)),
            ("mdx:other", ("mdx::",)))


def families(rows):
    """kernel time and launches per family and per step (a step = one launch of the training kernel)."""
    steps = sum(int(r["Calls"]) for r in rows if "photometric_train_kernel" in r["Name"]) or 1
    acc = {}
    for r in rows:
        fam = next((f for f, keys in FAMILIES if any(k in r["Name"] for k in keys)), "aten / other")
        t, n = acc.get(fam, (0.0, 0))
        acc[fam] = (t + float(r["TotalDurationNs"]), n + int(r["Calls"]))
    tot = sum(t for t, _ in acc.values())
    out = ["# per family, per step (%d steps profiled; under the profiler the two networks' kernels run one after the other)  total %.2f ms/step"
           % (steps, tot / steps / 1e6)]
    for fam, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        out.append("#   %-26s %8.3f ms/step %6.1f %% %8.1f calls/step" % (fam, t / steps / 1e6, 100 * t / tot, n / steps))
    return out


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
    lines += families(rows)
    open(dst, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:24]))


if __name__ == "__main__":
    main()
