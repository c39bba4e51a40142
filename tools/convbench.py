#!/usr/bin/env python3
"""Every distinct convolution of a training step (depth network + pose network of a BASELINE configuration), timed on its own:
forward, data gradient, weight gradient -- MIOpen through ATen, the package's find-db installed, maps in the step's layout
(channels-last by default).  Each number is GPU time per call from a replayed hipGraph of K calls.  FLOP/s and the bytes every
pass has to touch at least once (x, w, y) say which convolutions MIOpen runs far from both roofs.

    python tools/convbench.py [--bf16] [--planar] [--height 192 --width 640 --batch 12 --num-layers 18] [--json out.json]
"""
import argparse
import collections
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from netbench import graph_time  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=12)
    ap.add_argument("--height", type=int, default=192)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--num-layers", type=int, default=18)
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--planar", action="store_true")
    ap.add_argument("--json", default="")
    a = ap.parse_args()
    bench = importlib.import_module("bench")
    from model_train import trainer
    pkg.install_miopen_db(0)
    opt = bench.make_opt(a.batch, height=a.height, width=a.width, num_layers=a.num_layers, amp="bf16" if a.bf16 else "none")
    opt.channels_last, opt.graph, opt.miopen_find, opt.max_steps = ("none" if a.planar else "auto"), False, False, 0
    tr = trainer(opt)
    tr.setting.set_train()
    inputs = bench.one_batch(tr.setting, tr.device)
    seen = collections.OrderedDict()
    real = torch.nn.functional.conv2d

    def spy(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        two = lambda v: (v, v) if isinstance(v, int) else tuple(v)      # noqa: E731
        cl = x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        key = (tuple(x.shape), tuple(w.shape), two(stride), two(padding), x.dtype, cl, bool(x.requires_grad), b is not None)
        seen[key] = seen.get(key, 0) + 1
        return real(x, w, b, stride, padding, dilation, groups)
    torch.nn.functional.conv2d = spy
    try:
        tr._eager_step(inputs)
    finally:
        torch.nn.functional.conv2d = real
    torch.cuda.synchronize()
    rows = []
    for (xs, wsh, stride, pad, dt, cl, xgrad, has_b), count in seen.items():
        fmt = torch.channels_last if cl else torch.contiguous_format
        x = torch.randn(xs, device="cuda").to(dt).contiguous(memory_format=fmt)
        w = torch.randn(wsh, device="cuda").to(dt).contiguous(memory_format=fmt)
        y = real(x, w, None, stride, pad)
        gy = torch.randn_like(y)
        args = (list(stride), list(pad), [1, 1], False, [0, 0], 1)
        tf = graph_time(lambda: torch.ops.aten.convolution(x, w, None, *args))
        td = graph_time(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *args, [True, False, False])) if xgrad else 0.0
        tw = graph_time(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, *args, [False, True, False]))
        flop = 2.0 * y.numel() * wsh[1] * wsh[2] * wsh[3]
        nbytes = (x.numel() + y.numel() + w.numel()) * x.element_size()
        rows.append(dict(x=list(xs), w=list(wsh), stride=stride[0], pad=pad[0], count=count, channels_last=cl, fwd_us=tf, dgrad_us=td,
                         wgrad_us=tw, gflop=flop / 1e9, mbytes=nbytes / 1e6))
    rows.sort(key=lambda r: -(r["fwd_us"] + r["dgrad_us"] + r["wgrad_us"]) * r["count"])
    tot = sum((r["fwd_us"] + r["dgrad_us"] + r["wgrad_us"]) * r["count"] for r in rows)
    print("%-22s %-18s s p  n   fwd | dgrad | wgrad us   TFLOP/s fwd|dgrad|wgrad   floor us (flop|bytes)   share" % ("x", "w"))
    for r in rows:
        t3 = (r["fwd_us"], r["dgrad_us"], r["wgrad_us"])
        tf = ["%5.1f" % (r["gflop"] / t * 1e3) if t else "    -" for t in t3]          # GFLOP / us = 1e15 FLOP/s -> TFLOP/s
        peak = 2500.0 if a.bf16 else 157.0
        print("%-22s %-18s %d %d %2d  %6.1f | %6.1f | %6.1f   %s|%s|%s   %6.1f | %6.1f   %5.1f %%" % (
            "x".join(map(str, r["x"])), "x".join(map(str, r["w"])), r["stride"], r["pad"], r["count"], *t3, *tf,
            r["gflop"] / peak * 1e3, r["mbytes"] / 6.3, 100.0 * sum(t3) * r["count"] / tot))
    print("sum over the step's convolutions: %.3f ms (forward %.3f, data gradient %.3f, weight gradient %.3f)" % (
        tot / 1e3, sum(r["fwd_us"] * r["count"] for r in rows) / 1e3, sum(r["dgrad_us"] * r["count"] for r in rows) / 1e3,
        sum(r["wgrad_us"] * r["count"] for r in rows) / 1e3))
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
