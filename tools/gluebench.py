#!/usr/bin/env python3
"""Times the network glue kernels (csrc/glue.hip) on the shapes of the flagship step (B=12, 192x640, ResNet-18)
against the torch op sequences they replace, and prints achieved GB/s (bytes read + written / time).

    python tools/gluebench.py [--dtype bf16]
"""
import argparse
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
from mdx import functional as F  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "f32" else torch.bfloat16
    es = 4 if a.dtype == "f32" else 2
    B = 12
    # (C1, C2, h, w, elu, up) of the decoder's glue calls, deepest stage first
    calls = [(512, 0, 6, 20, False, False), (256, 256, 6, 20, True, True), (256, 0, 12, 40, True, False),
             (128, 128, 12, 40, True, True), (128, 0, 24, 80, True, False), (64, 64, 24, 80, True, True),
             (64, 0, 48, 160, True, False), (32, 64, 48, 160, True, True), (32, 0, 96, 320, True, False),
             (16, 0, 96, 320, True, True), (16, 0, 192, 640, True, False)]
    tot = [0.0, 0.0]
    for C1, C2, h, w, elu, up in calls:
        u = 2 if up else 1
        raw = torch.randn(B, C1, h, w, device="cuda").to(dt).requires_grad_(True)
        skip = torch.randn(B, C2, u * h, u * w, device="cuda").to(dt).requires_grad_(True) if C2 else None
        out = F.decoder_glue(raw, skip, elu=elu, upsample=up)
        gout = torch.randn_like(out)

        def ref():
            x = torch.nn.functional.elu(raw) if elu else raw
            if up:
                x = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
            if C2:
                x = torch.cat((x, skip), 1)
            return torch.nn.functional.pad(x, (1, 1, 1, 1), mode="reflect")

        def fwd():
            return F.decoder_glue(raw, skip, elu=elu, upsample=up)

        def both(f):
            o = f()
            torch.autograd.grad(o, [raw] + ([skip] if C2 else []), gout)

        t_f, t_fb, t_rf, t_rfb = timeit(fwd), timeit(lambda: both(fwd)), timeit(ref), timeit(lambda: both(ref))
        nbytes_f = (raw.numel() + (skip.numel() if C2 else 0) + out.numel()) * es
        nbytes_b = (out.numel() + 2 * raw.numel() + (skip.numel() if C2 else 0)) * es
        print("glue C1=%3d C2=%3d %3dx%3d up=%d  fwd %6.1f us (%5.0f GB/s)  bwd %6.1f us (%5.0f GB/s) | torch fwd %6.1f bwd %6.1f us"
              % (C1, C2, h, w, up, t_f, nbytes_f / t_f / 1e3, t_fb - t_f, nbytes_b / (t_fb - t_f) / 1e3, t_rf, t_rfb - t_rf))
        tot[0] += t_fb
        tot[1] += t_rfb
    x = torch.relu(torch.randn(B, 64, 96, 320, device="cuda")).to(dt).requires_grad_(True)
    y = F.maxpool3s2(x)
    gy = torch.randn_like(y)
    f1, f2 = (lambda: F.maxpool3s2(x)), (lambda: torch.nn.functional.max_pool2d(x, 3, 2, 1))
    t_f, t_r = timeit(f1), timeit(f2)
    t_fb = timeit(lambda: torch.autograd.grad(f1(), x, gy))
    t_rb = timeit(lambda: torch.autograd.grad(f2(), x, gy))
    print("maxpool 64ch 96x320: fwd %.1f us (%.0f GB/s) bwd %.1f us (%.0f GB/s) | torch fwd %.1f bwd %.1f us"
          % (t_f, (x.numel() + y.numel()) * es / t_f / 1e3, t_fb - t_f,
             (x.numel() * es + y.numel() * (es + 1)) / (t_fb - t_f) / 1e3, t_r, t_rb - t_r))
    print("decoder glue total fwd+bwd: %.0f us (torch ops: %.0f us)" % (tot[0], tot[1]))


if __name__ == "__main__":
    main()
