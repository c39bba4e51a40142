"""Entry-by-entry error of the disparity gradient (VERDICT r2 weak #1a,c): the one-launch training kernel (GPU, float32)
and the float32 CPU oracle, each against (i) the reference's own float32 autograd gradient (goldens) or each other and
(ii) the float64 evaluation of the same formulas with the same arg-min indices (oracle `make f64`).

On the "significant" entries (|g| >= 1e-2 max|g|) it prints the max-norm error, the worst and the 99 / 99.9 % quantiles of
the elementwise relative error, and the share of entries beyond 1e-4.  The float32-vs-float64 line is the noise floor: what
ANY float32 evaluation of this gradient (the reference's included) is uncertain by.
    python tools/diag_grad_elementwise.py > gpurun_out/grad_elementwise.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import gpu_util as G   # noqa: E402
import goldens         # noqa: E402
from oracle import oracle as orc   # noqa: E402
from test_gpu_parity import _synth   # noqa: E402
from test_gpu_train import _synth_images   # noqa: E402


def report(tag, a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    mx = np.abs(b).max() + 1e-300
    sig = np.abs(b) >= 1e-2 * mx
    e = np.abs(a - b)[sig] / np.abs(b)[sig]
    print("  %-34s max-norm %.2e | significant %5.1f%%: elementwise worst %.2e  p99.9 %.2e  p99 %.2e  median %.2e  share > 1e-4: %.3f%%"
          % (tag, np.abs(a - b).max() / mx, 100 * sig.mean(), e.max(), np.quantile(e, 0.999), np.quantile(e, 0.99),
             np.median(e), 100 * (e > 1e-4).mean()))


def oracle_case(name, maker, B, H, W, S, seed, nsc):
    colors, K, invK, Ts, rng = maker(B, H, W, S, seed=seed)
    disps_np = [rng.rand(B, 1, H >> s, W >> s).astype(np.float32) for s in range(nsc)]
    noises_np = [rng.randn(B, S, H, W).astype(np.float32) for _ in range(nsc)]
    P_ref = np.stack([orc.compose_projection(K, T) for T in Ts])
    srcs = [G.t(x) for x in colors[1:]]
    ident = G.F.identity_loss(G.t(colors[0]), srcs)
    n = B * H * W
    d1 = [G.t(x).requires_grad_(True) for x in disps_np]
    P1 = G.t(P_ref).requires_grad_(True)
    out = G.F.photometric_train(d1, P1, G.t(colors[0]), srcs, G.t(invK), ident, [G.t(x) for x in noises_np])
    (out["sums"].sum() / n).backward()
    print("%s  B=%d %dx%d S=%d" % (name, B, H, W, S))
    for s in range(nsc):
        ref = orc.photometric_fwd(disps_np[s], colors[0], colors[1:], invK, P_ref, noises_np[s], full=True)
        g32, _ = orc.photometric_bwd(disps_np[s], colors[0], colors[1:], invK, P_ref, ref["idx"], 1.0 / n)
        g64, _ = orc.photometric_bwd_f64(disps_np[s], colors[0], colors[1:], invK, P_ref, ref["idx"], 1.0 / n)
        gpu = d1[s].grad.cpu().numpy()
        print(" scale %d (%.0f%% auto-masked)" % (s, 100 * (ref["idx"] < S).mean()))
        report("GPU kernel vs float32 oracle", gpu, g32)
        report("GPU kernel vs float64 evaluation", gpu, g64)
        report("float32 oracle vs float64 (floor)", g32, g64)


def golden_case(name):
    c = goldens.Case(name)
    K = G.t(c["K"])
    Ts = {f: G.t(c.T(f)) for f in c.sources_ids}
    P = torch.stack([G.F.compose_projection(K, Ts[f]) for f in c.sources_ids])
    srcs = [G.t(c.color(f)) for f in c.sources_ids]
    n = c.B * c.H * c.W
    ident = G.F.identity_loss(G.t(c.color(0)), srcs) if c.automask else None
    disps = [G.t(c["disp_s%d" % s]).requires_grad_(True) for s in range(c.n_scales)]
    noises = [G.t(c["noise_s%d" % s]) for s in range(c.n_scales)] if c.automask else None
    out = G.F.photometric_train(disps, P, G.t(c.color(0)), srcs, G.t(c["inv_K"]), ident, noises, automask=c.automask)
    (out["sums"].sum() / (n * c.n_scales)).backward()
    print("golden %s (the reference's own float32 autograd gradient; photometric + smoothness in the golden, so the "
          "smoothness gradient is added)" % name)
    for s in range(c.n_scales):
        d2 = G.t(c["disp_s%d" % s]).requires_grad_(True)
        (1e-3 * G.F.smooth_loss(d2, G.t(c.color(0, s))) / (2 ** s) / c.n_scales).backward()
        report("scale %d GPU vs reference" % s, (disps[s].grad + d2.grad).cpu().numpy(), c["grad_disp_s%d" % s])


if __name__ == "__main__":
    for name in ("multi_64x160_b2", "mono_24x40_b2", "border_24x40_b2"):
        golden_case(name)
    oracle_case("image-like colours", _synth_images, 2, 192, 640, 2, 4321, 4)
    oracle_case("white-noise colours", _synth, 2, 192, 640, 2, 4323, 4)
