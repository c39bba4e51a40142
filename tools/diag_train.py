"""Diagnostic: gradient error of the one-launch training kernel and of the per-scale kernels against the oracle."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import gpu_util as G
from test_gpu_parity import _synth
from oracle import oracle as orc

B, H, W, S, nsc = 2, 192, 640, 2, 4
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4323
colors, K, invK, Ts, rng = _synth(B, H, W, S, seed=seed)
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
for s in range(nsc):
    ref = orc.photometric_fwd(disps_np[s], colors[0], colors[1:], invK, P_ref, noises_np[s], full=True)
    gd, gP = orc.photometric_bwd(disps_np[s], colors[0], colors[1:], invK, P_ref, ref["idx"], 1.0 / n)
    d2 = G.t(disps_np[s]).requires_grad_(True)
    P2 = G.t(P_ref).requires_grad_(True)
    o = G.F.photometric_scale(d2, P2, G.t(colors[0]), srcs, G.t(invK), ident, G.t(noises_np[s]))
    (o["sum"][0] / n).backward()
    a = d1[s].grad.cpu().numpy().astype(np.float64); b = d2.grad.cpu().numpy().astype(np.float64)
    sc = np.abs(gd).max()
    ea, eb = np.abs(a - gd) / sc, np.abs(b - gd) / sc
    ia = np.unravel_index(ea.argmax(), ea.shape)
    print("scale %d: max|g| %.3e  new-vs-oracle %.3e at %s (oracle %.4e new %.4e old %.4e)  old-vs-oracle %.3e  new-vs-old %.3e"
          % (s, sc, ea.max(), ia, gd[ia], a[ia], b[ia], eb.max(), np.abs(a - b).max() / sc))
    print("   count err>3e-5: new %d old %d ; rms new %.2e old %.2e" % ((ea > 3e-5).sum(), (eb > 3e-5).sum(), np.sqrt((ea**2).mean()), np.sqrt((eb**2).mean())))
