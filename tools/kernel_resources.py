#!/usr/bin/env python3
"""Registers / LDS / occupancy of the hand-written kernels as the shipped flags compile them (hipcc -S, no GPU needed):
    python tools/kernel_resources.py > profiles/r03_kernel_resources.txt
Occupancy: waves per SIMD = min(8, 512 // vgprs rounded up to 8, LDS limit 160 KB per CU / 4 SIMDs)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "digging-into-self-supervised-monocular-depth-estimation_amd")
sys.path.insert(0, PKG)
import build as B   # noqa: E402


def main():
    files = sys.argv[1:] or ["photo_train.hip", "photo_prologue.hip", "photo_fwd.hip", "imgproc.hip", "smooth.hip"]
    print("# kernel resources, flags of build.py (%s)" % " ".join(B.FLAGS))
    print("%-96s %5s %5s %6s %7s %6s %s" % ("kernel", "vgpr", "sgpr", "spill", "lds B", "waves", "threads/block"))
    for f in files:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            cmd = [B.HIPCC] + B.FLAGS + B.EXTRA_FLAGS.get(f, []) + ["-S", "--cuda-device-only", "-o", out, os.path.join(B.CSRC, f)]
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            text = open(out).read()
        demangle = subprocess.run(["c++filt"], input="\n".join(re.findall(r"\.name:\s+(\S+)", text)),
                                  capture_output=True, text=True).stdout.split("\n")
        names = re.findall(r"\.name:\s+(\S+)", text)
        pretty = dict(zip(names, demangle))
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
            blk = m.group(0)
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))   # noqa: E731
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            vg, lds, tpb = g("vgpr_count"), g("group_segment_fixed_size"), g("max_flat_workgroup_size")
            waves = min(8, 512 // max(8, (vg + 7) // 8 * 8))
            if lds:
                per_block_waves = max(1, tpb // 64)
                waves = min(waves, (160 * 1024 // lds) * per_block_waves // 4)
            short = pretty.get(name, name).replace("mdx::", "").split("(")[0]
            print("%-96s %5d %5d %6d %7d %6d %d" % (short[:96], vg, g("sgpr_count"), g("vgpr_spill_count"), lds, waves, tpb))


if __name__ == "__main__":
    main()
