"""GPU parity of the one-launch training kernel (csrc/photo_train.hip, mdx_photometric_train): every scale, forward and
gradient, against
  (1) the golden vectors recorded from the reference (tests/golden/*.npz): arg-min indices and to_optimise bit-exact,
      loss 1e-5, gradients 1e-4 (BASELINE.json north_star tolerance);
  (2) the CPU oracle on seeded inputs at BASELINE sizes, including configs[1]'s batch 12 and configs[3]'s 320x1024;
  (3) the per-scale kernels (photo_fwd.hip / photo_bwd.hip), which stay in the library as cross-checks.
"""
import numpy as np
import pytest
import torch

import goldens
from test_gpu_parity import _synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


@pytest.fixture(scope="module", params=goldens.CASES)
def case(request):
    return goldens.Case(request.param)


@pytest.mark.parametrize("rows", [0, 8, 5])
def test_train_kernel_vs_golden(G, case, rows):
    """compute_loss (processor.py:166-217) + autograd through the one-launch kernel, against the reference's."""
    c = case
    K = G.t(c["K"])
    Ts = {f: G.t(c.T(f)).requires_grad_(f != "s") for f in c.sources_ids}
    P = torch.stack([G.F.compose_projection(K, Ts[f]) for f in c.sources_ids])
    srcs = [G.t(c.color(f)) for f in c.sources_ids]
    n = c.B * c.H * c.W
    ident = G.F.identity_loss(G.t(c.color(0)), srcs) if c.automask else None
    disps = [G.t(c["disp_s%d" % s]).requires_grad_(True) for s in range(c.n_scales)]
    noises = [G.t(c["noise_s%d" % s]) for s in range(c.n_scales)] if c.automask else None
    out = G.F.photometric_train(disps, P, G.t(c.color(0)), srcs, G.t(c["inv_K"]), ident, noises,
                                automask=c.automask, need_depth=True, need_to_opt=True, rows_per_chunk=rows)
    G.assert_bitexact(out["depth"], c["depth_s0"], "depth s0")
    total = 0
    for s in range(c.n_scales):
        tgt = c["to_optimise_s%d" % s]
        G.assert_bitexact(out["to_opt"][s].reshape(tgt.shape), tgt, "to_optimise s%d" % s)
        if "idx_s%d" % s in c:
            assert (out["idx"][s].cpu().numpy() == c["idx_s%d" % s]).all(), "auto-mask indices s%d" % s
        G.assert_close(out["sums"][s].detach().cpu().numpy() / n, tgt.astype(np.float64).mean(), "mean s%d" % s, rel=1e-6)
        sm = G.F.smooth_loss(disps[s], G.t(c.color(0, s)))
        total = total + out["sums"][s] / n + 1e-3 * sm / (2 ** s)
    loss = total / c.n_scales
    loss.backward()
    G.assert_close(loss, c["loss"], "loss", rel=1e-5)
    for s in range(c.n_scales):
        G.assert_close(disps[s].grad, c["grad_disp_s%d" % s], "grad disp s%d" % s, elem=1e-4)
    for f in c.sources_ids:
        if f != "s":
            G.assert_close(Ts[f].grad, c["grad_T_%s" % f], "grad T %s" % f)


def _synth_images(B, H, W, S, seed):
    """_synth's cameras with image-like colours: low-pass random fields plus fine texture, quantised to uint8/255.
    White-noise images (what _synth makes) keep every SSIM window ill-conditioned: there a handful of pixels per
    image carry 1e-4 float32 rounding in ANY float32 evaluation of the gradient (the round-1 kernels and this one
    differ from the oracle at the same pixels, tools/diag_train.py), so gradients are checked on these images and the
    white-noise cases check what must be bit-exact."""
    from scipy.ndimage import gaussian_filter
    colors, K, invK, Ts, rng = _synth(B, H, W, S, seed=seed)
    base = gaussian_filter(rng.randn(B, 3, H + 16, W + 16), (0, 0, 6, 6)) * 8.0
    out = []
    for k in range(S + 1):
        dy, dx = rng.randint(0, 9, size=2)
        img = 0.5 + 0.35 * base[:, :, dy:dy + H, dx:dx + W] + 0.08 * gaussian_filter(rng.randn(B, 3, H, W), (0, 0, 1, 1)) * 3
        out.append((np.clip(np.round(img * 255), 0, 255).astype(np.float32) / np.float32(255.0)).astype(np.float32))
    return out, K, invK, Ts, rng


def _oracle_case(G, B, H, W, S, seed, nscales=4, automask=True, grads=True, rows=0, white_noise=False, grad_rel=1e-4,
                 pre=False):
    """pre=True: the step's own path -- mdx_photometric_prologue (injected noise) feeds mdx_photometric_train_pre, i.e. the
    <S, grad, PRE> instantiation bench.py times -- instead of the ident + noise form."""
    from oracle import oracle as orc
    colors, K, invK, Ts, rng = (_synth if white_noise else _synth_images)(B, H, W, S, seed=seed)
    hw = [(H >> s, W >> s) if H % 8 == 0 and W % 8 == 0 else (H, W) for s in range(nscales)]
    disps_np = [rng.rand(B, 1, h, w).astype(np.float32) for h, w in hw]
    noises_np = [rng.randn(B, S, H, W).astype(np.float32) for _ in range(nscales)] if automask else None
    P_ref = np.stack([orc.compose_projection(K, T) for T in Ts])
    srcs = [G.t(x) for x in colors[1:]]
    disps = [G.t(x).requires_grad_(True) for x in disps_np]
    Pt = G.t(P_ref).requires_grad_(True)
    if pre:
        prologue = G.F.photometric_prologue(G.t(colors[0]), srcs, nscales, automask=automask,
                                            noises=[G.t(x) for x in noises_np] if automask else None)
        out = G.F.photometric_train(disps, Pt, G.t(colors[0]), srcs, G.t(invK), automask=automask, need_depth=True,
                                    need_to_opt=True, rows_per_chunk=rows, pre=prologue)
    else:
        ident = G.F.identity_loss(G.t(colors[0]), srcs) if automask else None
        out = G.F.photometric_train(disps, Pt, G.t(colors[0]), srcs, G.t(invK), ident,
                                    [G.t(x) for x in noises_np] if automask else None, automask=automask,
                                    need_depth=True, need_to_opt=True, rows_per_chunk=rows)
    n = B * H * W
    (out["sums"].sum() / n).backward()
    gP_ref = 0
    for s in range(nscales):
        ref = orc.photometric_fwd(disps_np[s], colors[0], colors[1:], invK, P_ref,
                                  noises_np[s] if automask else None, automask=automask, full=True)
        if s == 0:
            G.assert_bitexact(out["depth"], ref["depth"], "depth")
        G.assert_bitexact(out["to_opt"][s], ref["to_opt"], "to_opt s%d" % s)
        assert (out["idx"][s].cpu().numpy() == ref["idx"]).all(), "auto-mask indices s%d" % s
        G.assert_close(out["sums"][s:s + 1], np.array([ref["sum"]]), "sum s%d" % s, rel=1e-6)
        if grads:
            gd, gP = orc.photometric_bwd(disps_np[s], colors[0], colors[1:], invK, P_ref, ref["idx"], 1.0 / n,
                                         automask=automask)
            G.assert_close(disps[s].grad, gd, "grad disp s%d" % s, rel=grad_rel, elem=grad_rel)
            gP_ref = gP_ref + gP
    if grads:
        G.assert_close(Pt.grad, gP_ref, "grad P", rel=grad_rel)


@pytest.mark.parametrize("B,H,W,S,nscales,automask", [
    (2, 192, 640, 2, 4, True),      # configs[1] tile shape, mono
    (1, 192, 640, 3, 4, True),      # configs[4]: mono + stereo
    (1, 320, 1024, 2, 2, True),     # configs[3] resolution
    (1, 100, 150, 2, 1, True),      # ragged: partial strips, H not a multiple of the chunk
    (1, 192, 640, 2, 2, False),     # use_automasking = False
    (3, 64, 68, 1, 4, True),        # single source frame, W just over one strip
    (1, 192, 640, 4, 1, True),      # MDX_MAX_SRC
    (2, 64, 160, 3, 2, False),      # S = 3 (the LOW form: accumulators / coefficient rows in LDS, (u, v) ring) without auto-masking
])
def test_train_kernel_vs_oracle_full_size(G, B, H, W, S, nscales, automask):
    _oracle_case(G, B, H, W, S, seed=4321 + S, nscales=nscales, automask=automask)


def test_train_kernel_vs_oracle_batch12(G):
    """configs[1] as bench.py runs it (12 x 192 x 640, S = 2, all four scales): indices, to_optimise, sums AND the
    gradients -- d(P) is then the fixed-order sum over the 12 images' work items (train_finish_kernel)."""
    _oracle_case(G, 12, 192, 640, 2, seed=77, nscales=4, grads=True)


@pytest.mark.parametrize("B,H,W,S,nscales", [
    (12, 192, 640, 2, 4),      # configs[1]: <2, grad, PRE> at the bench's batch -- the prologue and train_finish_kernel's fixed-order
                               # d(P) sum over 12 images' items, looked at together, against the oracle directly
    (2, 192, 640, 3, 4),       # configs[4]: <3, grad, PRE>
    (1, 320, 1024, 2, 4),      # configs[3]'s resolution (its own chunk schedule)
    (1, 100, 150, 2, 1),       # ragged
])
def test_train_kernel_with_prologue_vs_oracle(G, B, H, W, S, nscales):
    _oracle_case(G, B, H, W, S, seed=177 + S, nscales=nscales, grads=True, pre=True)


def test_train_kernel_vs_oracle_white_noise(G):
    """White-noise colours (every window ill-conditioned): the per-pixel tensors and indices stay bit-exact; the
    gradients, where any float32 evaluation carries ~1e-4 of rounding (tools/diag_grad_elementwise.py),
    stay within 5e-4 of the oracle's in the max norm."""
    _oracle_case(G, 2, 192, 640, 2, seed=4323, nscales=4, grads=True, white_noise=True, grad_rel=5e-4)


@pytest.mark.parametrize("B,H,W,S,nscales,automask", [
    (2, 192, 640, 2, 4, True),
    (1, 192, 640, 3, 4, True),
    (1, 100, 150, 2, 1, True),
    (1, 192, 640, 2, 2, False),
    (3, 64, 68, 1, 4, True),
])
def test_forward_only_form_equals_training_form(G, B, H, W, S, nscales, automask):
    """Under torch.no_grad() (validation, model_train.py:75-79) the same launch runs without its gradient phase: loss
    sums, indices, to_optimise and depth are those of the training form, bit for bit (and so the oracle's)."""
    colors, K, invK, Ts, rng = _synth_images(B, H, W, S, seed=99 + S)
    hw = [(H >> s, W >> s) if H % 8 == 0 and W % 8 == 0 else (H, W) for s in range(nscales)]
    disps_np = [rng.rand(B, 1, h, w).astype(np.float32) for h, w in hw]
    noises = [G.t(rng.randn(B, S, H, W).astype(np.float32)) for _ in range(nscales)] if automask else None
    Kt = G.t(K)
    P = torch.stack([G.F.compose_projection(Kt, G.t(T)) for T in Ts])
    srcs = [G.t(x) for x in colors[1:]]
    ident = G.F.identity_loss(G.t(colors[0]), srcs) if automask else None
    args = (P, G.t(colors[0]), srcs, G.t(invK), ident, noises)
    kw = dict(automask=automask, need_depth=True, need_to_opt=True)
    a = G.F.photometric_train([G.t(x).requires_grad_(True) for x in disps_np], *args, **kw)
    with torch.no_grad():
        b = G.F.photometric_train([G.t(x).requires_grad_(True) for x in disps_np], *args, **kw)
    c = G.F.photometric_train([G.t(x) for x in disps_np], *args, **kw)        # nothing requires a gradient
    for o in (b, c):
        assert not o["sums"].requires_grad
        assert torch.equal(o["sums"], a["sums"].detach()) and torch.equal(o["depth"], a["depth"])
        for s in range(nscales):
            assert torch.equal(o["idx"][s], a["idx"][s]) and torch.equal(o["to_opt"][s], a["to_opt"][s])


def test_train_kernel_matches_per_scale_kernels(G):
    """The per-scale forward + coefficient backward (what round 1 trained with) give the same numbers."""
    B, H, W, S = 2, 192, 640, 2
    colors, K, invK, Ts, rng = _synth_images(B, H, W, S, seed=5)
    srcs = [G.t(x) for x in colors[1:]]
    Kt = G.t(K)
    P = torch.stack([G.F.compose_projection(Kt, G.t(T)) for T in Ts])
    ident = G.F.identity_loss(G.t(colors[0]), srcs)
    disps_np = [rng.rand(B, 1, H >> s, W >> s).astype(np.float32) for s in range(4)]
    noises = [G.t(rng.randn(B, S, H, W).astype(np.float32)) for _ in range(4)]
    n = B * H * W
    d1 = [G.t(x).requires_grad_(True) for x in disps_np]
    P1 = P.clone().requires_grad_(True)
    out = G.F.photometric_train(d1, P1, G.t(colors[0]), srcs, G.t(invK), ident, noises)
    (out["sums"].sum() / n).backward()
    d2 = [G.t(x).requires_grad_(True) for x in disps_np]
    P2 = P.clone().requires_grad_(True)
    tot = 0
    for s in range(4):
        o = G.F.photometric_scale(d2[s], P2, G.t(colors[0]), srcs, G.t(invK), ident, noises[s])
        assert torch.equal(o["idx"], out["idx"][s])
        G.assert_close(out["sums"][s:s + 1], o["sum"].detach().cpu().numpy(), "sum s%d" % s, rel=1e-6)
        tot = tot + o["sum"][0]
    (tot / n).backward()
    for s in range(4):
        G.assert_close(d1[s].grad, d2[s].grad.cpu().numpy(), "grad disp s%d" % s, rel=1e-4)
    G.assert_close(P1.grad, P2.grad.cpu().numpy(), "grad P", rel=1e-4)


def test_smooth_loss_multi_vs_per_scale_and_golden(G):
    """mdx_smooth_loss_multi (every scale, each pass launched once) against the per-scale op and the reference's values."""
    for name in ("mono_24x40_b2", "multi_64x160_b2"):
        c = goldens.Case(name)
        d1 = [G.t(c["disp_s%d" % s]).requires_grad_(True) for s in range(c.n_scales)]
        d2 = [G.t(c["disp_s%d" % s]).requires_grad_(True) for s in range(c.n_scales)]
        cols = [G.t(c.color(0, s)) for s in range(c.n_scales)]
        multi = G.F.smooth_loss_multi(d1, cols)
        wts = torch.tensor([1.0 / (2 ** s) for s in range(c.n_scales)], device=G.DEV)
        (multi * wts).sum().backward()
        tot = 0
        for s in range(c.n_scales):
            one = G.F.smooth_loss(d2[s], cols[s])
            G.assert_close(multi[s], c["smooth_s%d" % s], "smooth s%d vs reference" % s)
            G.assert_close(multi[s], one.detach().cpu().numpy(), "smooth s%d vs per-scale op" % s, rel=1e-6)
            tot = tot + one * wts[s]
        tot.backward()
        for s in range(c.n_scales):
            G.assert_close(d1[s].grad, d2[s].grad.cpu().numpy(), "smooth grad s%d" % s, rel=1e-6)


def test_train_kernel_projection_per_scale(G):
    """posecnn-style: a different projection for every scale (processor.py:153-157) -- against the per-scale kernels."""
    B, H, W, S = 2, 64, 96, 2
    colors, K, invK, Ts, rng = _synth_images(B, H, W, S, seed=11)
    srcs = [G.t(x) for x in colors[1:]]
    Kt = G.t(K)
    ident = G.F.identity_loss(G.t(colors[0]), srcs)
    disps_np = [rng.rand(B, 1, H >> s, W >> s).astype(np.float32) for s in range(3)]
    noises = [G.t(rng.randn(B, S, H, W).astype(np.float32)) for _ in range(3)]
    Ps_np = []
    for s in range(3):
        Tsc = [T.copy() for T in Ts]
        for T in Tsc:
            T[:, :3, 3] *= (1.0 + 0.5 * s)
        Ps_np.append(torch.stack([G.F.compose_projection(Kt, G.t(T)) for T in Tsc]))
    n = B * H * W
    d1 = [G.t(x).requires_grad_(True) for x in disps_np]
    P1 = [p.clone().requires_grad_(True) for p in Ps_np]
    out = G.F.photometric_train(d1, P1, G.t(colors[0]), srcs, G.t(invK), ident, noises)
    (out["sums"] * torch.tensor([1.0, 0.5, 0.25], device=G.DEV)).sum().div(n).backward()
    d2 = [G.t(x).requires_grad_(True) for x in disps_np]
    P2 = [p.clone().requires_grad_(True) for p in Ps_np]
    tot = 0
    for s in range(3):
        o = G.F.photometric_scale(d2[s], P2[s], G.t(colors[0]), srcs, G.t(invK), ident, noises[s])
        assert torch.equal(o["idx"], out["idx"][s])
        tot = tot + o["sum"][0] * (0.5 ** s)
    (tot / n).backward()
    for s in range(3):
        G.assert_close(d1[s].grad, d2[s].grad.cpu().numpy(), "grad disp s%d" % s)
        G.assert_close(P1[s].grad, P2[s].grad.cpu().numpy(), "grad P s%d" % s)


@pytest.mark.parametrize("B,H,W,S", [
    (12, 192, 640, 2),      # BASELINE configs[1]: the bench's batch
    (12, 192, 640, 3),      # configs[4]: mono + stereo (the LOW form)
    (8, 320, 1024, 2),      # configs[3]
])
def test_step_path_properties_at_baseline_sizes(G, B, H, W, S):
    """Size-independent properties of the step's own path (prologue -> training kernel <S, grad, PRE> -> finishing pass) at the
    BASELINE batch sizes, where the oracle would take minutes: (1) two launches on the same inputs agree bit for bit in every
    output, gradients included (fixed summation orders, no atomics on floats); (2) an image's per-pixel results do not depend on
    the batch it sits in -- images 0 and B-1 run alone give the bits they got inside the batch (chunk schedule, work-item
    numbering and XCD grouping all differ between the two launches); (3) every scale's loss sum is the float64 sum of its
    to_optimise map to 1e-6."""
    colors, K, invK, Ts, rng = _synth_images(B, H, W, S, seed=97 + S)
    from oracle import oracle as orc
    nscales = 4
    disps_np = [rng.rand(B, 1, H >> s, W >> s).astype(np.float32) for s in range(nscales)]
    noises_np = [rng.randn(B, S, H, W).astype(np.float32) for _ in range(nscales)]
    P = np.stack([orc.compose_projection(K, T) for T in Ts])

    def run(sel):
        srcs = [G.t(x[sel]) for x in colors[1:]]
        disps = [G.t(x[sel]).requires_grad_(True) for x in disps_np]
        Pt = G.t(np.ascontiguousarray(P[:, sel])).requires_grad_(True)
        pre = G.F.photometric_prologue(G.t(colors[0][sel]), srcs, nscales, automask=True, noises=[G.t(x[sel]) for x in noises_np])
        out = G.F.photometric_train(disps, Pt, G.t(colors[0][sel]), srcs, G.t(invK[sel]), automask=True, need_depth=True,
                                    need_to_opt=True, pre=pre)
        out["sums"].sum().backward()
        return out, [d.grad for d in disps], Pt.grad

    full = slice(0, B)
    a, ga, gPa = run(full)
    b, gb, gPb = run(full)
    for s in range(nscales):
        assert torch.equal(a["idx"][s], b["idx"][s]) and torch.equal(a["to_opt"][s], b["to_opt"][s]), "determinism s%d" % s
        assert torch.equal(ga[s], gb[s]), "determinism of the disparity gradient s%d" % s
        tot = float(a["to_opt"][s].double().sum())
        assert abs(float(a["sums"][s].detach()) - tot) <= 1e-6 * abs(tot), "loss sum s%d" % s
    assert torch.equal(a["sums"], b["sums"]) and torch.equal(gPa, gPb) and torch.equal(a["depth"], b["depth"])
    for n in (0, B - 1):
        one, g1, _ = run(slice(n, n + 1))
        assert torch.equal(one["depth"][0], a["depth"][n])
        for s in range(nscales):
            assert torch.equal(one["idx"][s][0], a["idx"][s][n]), "image %d alone, idx s%d" % (n, s)
            assert torch.equal(one["to_opt"][s][0], a["to_opt"][s][n]), "image %d alone, to_optimise s%d" % (n, s)
            assert torch.equal(g1[s][0], ga[s][n]), "image %d alone, disparity gradient s%d" % (n, s)
