#!/usr/bin/env python3
"""Per-shape times of the hand-written network kernels between the convolutions (batch norm + add + ReLU, decoder glue,
max-pool), planar (csrc/norm.hip, glue.hip) against channels-last (csrc/norm_nhwc.hip, glue_nhwc.hip), on the maps of the
BASELINE configurations.  Each measurement captures K calls in a hipGraph and replays it: GPU time per call INCLUDING the
gaps between the launches of one call (what a step pays), not host time.  Bytes are the algorithmic ones (every map once
per pass that has to touch it: forward x [+ res] -> y; backward dy, y, x -> dx [+ dres]), so GB/s is comparable between
the layouts whatever their launch count.

    python tools/netbench.py [--bf16] [--config 1|3] [--json out.json]
"""
import argparse
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
from mdx import functional as F  # noqa: E402

K = 10


def graph_time(fn, reps=20):
    """us per call of fn, replaying a captured graph of K calls."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(K):
            fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (reps * K)


def fmt(cl):
    return torch.channels_last if cl else torch.contiguous_format


def bn_case(B, C, H, W, has_res, groups, dt, cl, two_grads=False):
    x = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=fmt(cl)).requires_grad_(True)
    res = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=fmt(cl)).requires_grad_(True) if has_res else None
    w, b = torch.ones(C, device="cuda", requires_grad=True), torch.zeros(C, device="cuda", requires_grad=True)
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    gy = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=fmt(cl))
    ins = [x, w, b] + ([res] if has_res else [])
    n = x.numel() * x.element_size()

    def f():
        return F.bn_act(x, w, b, rm, rv, 1e-5, 0.1, residual=res, relu=True, groups=groups)
    tf = graph_time(f)
    # backward = (forward + backward) - forward: both inside the capture (a backward of a graph built outside the capture would
    # run on the forward's stream, not on the capturing one)
    tb = graph_time(lambda: torch.autograd.grad(f(), ins, gy)) - tf
    bytes_f = n * (2 + (1 if has_res else 0))
    bytes_b = n * (4 + (1 if has_res else 0))
    return tf, tb, bytes_f, bytes_b


def glue_case(B, C1, C2, h, w, up, elu, dt, cl):
    u = 2 if up else 1
    raw = torch.randn(B, C1, h, w, device="cuda").to(dt).contiguous(memory_format=fmt(cl)).requires_grad_(True)
    skip = torch.randn(B, C2, u * h, u * w, device="cuda").to(dt).contiguous(memory_format=fmt(cl)).requires_grad_(True) if C2 else None
    bias = torch.randn(C1, device="cuda", requires_grad=True) if elu else None

    def f():
        return F.decoder_glue(raw, skip, elu=elu, upsample=up, bias=bias)
    tf = graph_time(f)
    with torch.no_grad():
        out = f()            # (no autograd nodes outside the capture: they would be tied to the default stream)
    gout = torch.randn_like(out)
    ins = [raw] + ([skip] if C2 else []) + ([bias] if elu else [])
    tb = graph_time(lambda: torch.autograd.grad(f(), ins, gout)) - tf
    es = raw.element_size()
    n_in = (raw.numel() + (skip.numel() if C2 else 0)) * es
    n_out = out.numel() * es
    return tf, tb, n_in + n_out, n_out + n_in + raw.numel() * es


def pool_case(B, C, H, W, dt, cl):
    x = torch.randn(B, C, H, W, device="cuda").to(dt).contiguous(memory_format=fmt(cl)).requires_grad_(True)

    def f():
        return F.maxpool3s2(x)
    tf = graph_time(f)
    with torch.no_grad():
        gy = torch.randn_like(f())
    tb = graph_time(lambda: torch.autograd.grad(f(), [x], gy)) - tf
    es = x.element_size()
    return tf, tb, x.numel() * es + gy.numel() * (es + 1), gy.numel() * (es + 1) + x.numel() * es


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--config", type=int, default=1, choices=[1, 3])
    ap.add_argument("--json", type=str, default="")
    ap.add_argument("--only", type=str, default="", help="bn | glue | pool: only this family")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.bf16 else torch.float32
    if a.config == 1:       # ResNet-18, 192x640, batch 12 (depth) / 2 x 12 (both pose pairs in one batch)
        B, H0, W0, ch = 12, 192, 640, (64, 64, 128, 256, 512)
    else:                   # ResNet-50, 320x1024, batch 8
        B, H0, W0, ch = 8, 320, 1024, (64, 256, 512, 1024, 2048)
    bn_shapes = [(B, ch[0], H0 // 2, W0 // 2, False, 1), (2 * B, ch[0], H0 // 2, W0 // 2, False, 2)]
    for i, c in enumerate(ch[1:]):
        hh, ww = H0 // (4 << i), W0 // (4 << i)
        bn_shapes += [(B, c, hh, ww, False, 1), (B, c, hh, ww, True, 1), (2 * B, c, hh, ww, True, 2)]
        if a.config == 3:
            bn_shapes += [(B, c // 4, hh, ww, False, 1)]
    rows = []
    print("%-46s %21s %21s" % ("op / shape", "planar fwd | bwd us (GB/s)", "channels-last fwd | bwd us (GB/s)"))
    for (b, c, hh, ww, res, g) in (bn_shapes if a.only in ("", "bn") else []):
        r = {}
        for cl in (False, True):
            r[cl] = bn_case(b, c, hh, ww, res, g, dt, cl)
        name = "bn_act B=%d C=%d %dx%d res=%d groups=%d" % (b, c, hh, ww, res, g)
        rows.append((name, r))
    dec = (16, 32, 64, 128, 256)
    glue_shapes = [(B, ch[4], 0, H0 // 32, W0 // 32, False, False)]
    for i in (4, 3, 2, 1, 0):
        hh, ww = H0 // (2 << i), W0 // (2 << i)
        glue_shapes += [(B, dec[i], ch[i - 1] if i > 0 else 0, hh, ww, True, True)]
        glue_shapes += [(B, dec[i], 0, 2 * hh, 2 * ww, False, True)]
    for (b, c1, c2, hh, ww, up, elu) in (glue_shapes if a.only in ("", "glue") else []):
        r = {}
        for cl in (False, True):
            r[cl] = glue_case(b, c1, c2, hh, ww, up, elu, dt, cl)
        rows.append(("decoder_glue B=%d C1=%d C2=%d %dx%d up=%d" % (b, c1, c2, hh, ww, up), r))
    for b in ((B, 2 * B) if a.only in ("", "pool") else ()):
        r = {}
        for cl in (False, True):
            r[cl] = pool_case(b, 64, H0 // 2, W0 // 2, dt, cl)
        rows.append(("maxpool3s2 B=%d C=64 %dx%d" % (b, H0 // 2, W0 // 2), r))
    out = []
    tot = {False: [0.0, 0.0], True: [0.0, 0.0]}
    for name, r in rows:
        cells = []
        for cl in (False, True):
            tf, tb, bf, bb = r[cl]
            cells.append("%6.1f (%4.0f) | %6.1f (%4.0f)" % (tf, bf / tf / 1e3, tb, bb / tb / 1e3))
            tot[cl][0] += tf
            tot[cl][1] += tb
            out.append({"op": name, "layout": "nhwc" if cl else "nchw", "dtype": str(dt), "fwd_us": tf, "bwd_us": tb,
                        "fwd_bytes": bf, "bwd_bytes": bb, "fwd_frac_hbm": bf / tf / 1e3 / 8000.0,
                        "bwd_frac_hbm": bb / tb / 1e3 / 8000.0})
        print("%-46s %s    %s" % (name, cells[0], cells[1]))
    print("sum over the listed shapes: planar %.0f + %.0f us, channels-last %.0f + %.0f us" % (tot[False][0], tot[False][1], tot[True][0], tot[True][1]))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
