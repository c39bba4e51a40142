"""Diagnostic: gradient error against the reference-made goldens, per case and scale, for the one-launch training
kernel and for the per-scale coefficient path (max |err| / max |ref|; the tests allow 1e-4)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import gpu_util as G
import goldens


def err(a, b):
    a = a.detach().cpu().numpy().astype(np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


for name in goldens.CASES:
    c = goldens.Case(name)
    n = c.B * c.H * c.W
    res = {}
    for path in ("train", "scale"):
        K = G.t(c["K"])
        Ts = {f: G.t(c.T(f)).requires_grad_(f != "s") for f in c.sources_ids}
        P = torch.stack([G.F.compose_projection(K, Ts[f]) for f in c.sources_ids])
        srcs = [G.t(c.color(f)) for f in c.sources_ids]
        ident = G.F.identity_loss(G.t(c.color(0)), srcs) if c.automask else None
        disps = [G.t(c["disp_s%d" % s]).requires_grad_(True) for s in range(c.n_scales)]
        noises = [G.t(c["noise_s%d" % s]) for s in range(c.n_scales)] if c.automask else None
        if path == "train":
            out = G.F.photometric_train(disps, P, G.t(c.color(0)), srcs, G.t(c["inv_K"]), ident, noises, automask=c.automask)
            sums = [out["sums"][s] for s in range(c.n_scales)]
        else:
            sums = [G.F.photometric_scale(disps[s], P, G.t(c.color(0)), srcs, G.t(c["inv_K"]), ident,
                                          noises[s] if c.automask else None, automask=c.automask)["sum"][0]
                    for s in range(c.n_scales)]
        total = 0
        for s in range(c.n_scales):
            sm = G.F.smooth_loss(disps[s], G.t(c.color(0, s)))
            total = total + sums[s] / n + 1e-3 * sm / (2 ** s)
        (total / c.n_scales).backward()
        res[path] = ([err(disps[s].grad, c["grad_disp_s%d" % s]) for s in range(c.n_scales)],
                     [err(Ts[f].grad, c["grad_T_%s" % f]) for f in c.sources_ids if f != "s"])
    print("%-22s train: disp %s  T %s" % (name, " ".join("%.1e" % e for e in res["train"][0]), " ".join("%.1e" % e for e in res["train"][1])))
    print("%-22s scale: disp %s  T %s" % ("", " ".join("%.1e" % e for e in res["scale"][0]), " ".join("%.1e" % e for e in res["scale"][1])))
