#!/usr/bin/env python3
"""Micro-benchmark of the image-preparation kernels alone (csrc/imgproc.hip), for rocprofv3 --kernel-trace / --pmc.

    python tools/imgbench.py [--n 32] [--reps 20] [--out 192x640] [--in 375x1242] [--jitter]
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
from mdx import imgproc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=32)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--out", type=str, default="192x640")
    ap.add_argument("--in", dest="inp", type=str, default="375x1242")
    ap.add_argument("--jitter", action="store_true")
    ap.add_argument("--empty", action="store_true", help="with --jitter: the empty chain (ToTensor alone) for every image")
    a = ap.parse_args()
    oh, ow = (int(v) for v in a.out.split("x"))
    h, w = (int(v) for v in a.inp.split("x"))
    g = torch.Generator().manual_seed(0)
    lo = torch.rand(a.n, 3, h // 8, w // 8, generator=g)
    img = torch.nn.functional.interpolate(lo, size=(h, w), mode="bilinear", align_corners=False)
    src = (img * 255).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous().cuda()
    plans = imgproc.plan_cache("cuda:0")
    sizes, flips = [(h, w)] * a.n, [bool(i % 2) for i in range(a.n)]
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    def run():
        return imgproc.resize_lanczos(plans, src, sizes, flips, (oh, ow), want_u8=True)
    u8 = run()[0]
    for _ in range(3):
        run()
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(a.reps):
        run()
    e1.record()
    e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / a.reps
    alg = a.n * (h * w * 3 + oh * ow * 3 * 5)            # source bytes once + uint8 and float32 outputs
    print("resize %d x (%dx%d -> %dx%d): %.1f us per call, %.2f us per image; algorithmic %.1f MB -> %.0f GB/s"
          % (a.n, h, w, oh, ow, us, us / a.n, alg / 1e6, alg / us / 1e3))
    if a.jitter:
        params = [([4, 4, 4, 4], 1.0, 1.0, 1.0, 0) if a.empty else ([2, 0, 3, 1], 1.1, 0.9, 1.15, -14)] * a.n
        out = torch.empty(a.n, 3, oh, ow, device="cuda")
        for _ in range(3):
            imgproc.color_jitter(u8, params, out)
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(a.reps):
            imgproc.color_jitter(u8, params, out)
        e1.record()
        e1.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / a.reps
        print("jitter %d x %dx%d: %.1f us per call, %.2f us per image" % (a.n, oh, ow, us, us / a.n))


if __name__ == "__main__":
    main()
