#!/usr/bin/env python3
"""Times bn_act (csrc/norm.hip) against torch's batch_norm (MIOpen) + add + relu on the ResNet-18 shapes of the
flagship step (B=12, 192x640), forward and backward, kernel time by HIP events."""
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
    dt = torch.bfloat16 if "--bf16" in sys.argv else torch.float32
    B = 12
    for C, H, W, has_res in [(64, 96, 320, False), (64, 48, 160, False), (64, 48, 160, True), (128, 24, 80, True),
                             (256, 12, 40, True), (512, 6, 20, True)]:
        x = torch.randn(B, C, H, W, device="cuda").to(dt).requires_grad_(True)
        res = torch.randn(B, C, H, W, device="cuda").to(dt).requires_grad_(True) if has_res else None
        w, b = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        gy = torch.randn(B, C, H, W, device="cuda").to(dt)
        ins = [x, w, b] + ([res] if has_res else [])

        def mine():
            return F.bn_act(x, w, b, rm, rv, 1e-5, 0.1, residual=res, relu=True)

        def ref():
            o = torch.nn.functional.batch_norm(x, rm, rv, w, b, True, 0.1, 1e-5)
            if has_res:
                o = o + res
            return torch.relu(o)
        t = [timeit(f) for f in (mine, ref)]
        tb = [timeit(lambda: torch.autograd.grad(f(), ins, gy)) for f in (mine, ref)]
        n = x.numel() * x.element_size()
        print("C=%3d %3dx%3d res=%d  fwd %6.1f us | torch %6.1f us   fwd+bwd %6.1f us | torch %6.1f us   (event time "
              "incl. Python; kernel times: rocprofv3 --kernel-trace)" % (C, H, W, has_res, t[0], t[1], tb[0], tb[1]))


if __name__ == "__main__":
    main()
