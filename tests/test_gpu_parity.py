"""GPU parity tests (MI355X): the HIP path, called through the C-ABI, against
  (1) the golden vectors recorded from the reference itself (tests/golden/*.npz), and
  (2) the CPU oracle on seeded inputs at BASELINE sizes.
Bit-exact for per-pixel tensors and auto-mask indices; 1e-4 rel (north_star) for scalar
reductions and gradients.
"""
import numpy as np
import pytest
import torch

import goldens

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


@pytest.fixture(scope="module", params=goldens.CASES)
def case(request):
    return goldens.Case(request.param)


def _P(G, c):
    K = G.t(c["K"])
    return torch.stack([G.F.compose_projection(K, G.t(c.T(f))) for f in c.sources_ids])


def test_library_loaded(G):
    from mdx import lib, LIB_PATH
    assert lib().mdx_version() == 510
    assert LIB_PATH.endswith("libmdx_hip.so")


def test_compose_projection(G):
    for name in goldens.FULL_CASES:
        c = goldens.Case(name)
        P = _P(G, c)
        for i, f in enumerate(c.sources_ids):
            G.assert_bitexact(P[i], c["P_%s" % f], "P %s" % f)


def test_fused_forward_vs_golden(G, case):
    c = case
    P = _P(G, c)
    srcs = [G.t(c.color(f)) for f in c.sources_ids]
    full = ("warp_%s_s0" % c.sources_ids[0]) in c
    n = c.B * c.H * c.W
    for s in range(c.n_scales):
        out, _, _ = G.run_scale(c, s, srcs, P, want_grad=False, need_to_opt=True, need_depth=True,
                                need_warp=True, need_reproj=True)
        G.assert_bitexact(out["depth"], c["depth_s%d" % s], "depth s%d" % s)
        if full:
            for i, f in enumerate(c.sources_ids):
                G.assert_bitexact(out["warp"][i], c["warp_%s_s%d" % (f, s)], "warp %s s%d" % (f, s))
            if "combined_s%d" % s in c and not c.automask:
                G.assert_bitexact(out["reproj"], c["combined_s%d" % s], "reproj s%d" % s)
            if "combined_s%d" % s in c and c.automask:
                G.assert_bitexact(out["reproj"], c["combined_s%d" % s][:, c.S:], "reproj s%d" % s)
        tgt = c["to_optimise_s%d" % s]
        G.assert_bitexact(out["to_opt"].reshape(tgt.shape), tgt, "to_optimise s%d" % s)
        if "idx_s%d" % s in c:
            assert (out["idx"].cpu().numpy() == c["idx_s%d" % s]).all(), "auto-mask indices s%d" % s
        G.assert_close(out["sum"].cpu().numpy()[0] / n, tgt.astype(np.float64).mean(), "mean s%d" % s, rel=1e-6)


@pytest.mark.parametrize("bwd_path", ["coef", "saved_warp", "rewarp"])
def test_fused_loss_and_grads_vs_golden(G, case, bwd_path):
    """Whole loss of compute_loss (processor.py:166-217) + autograd, against the reference's -- through each of
    the three backward kernels (coefficient maps = what training runs; saved warped colours; full re-warp)."""
    c = case
    keep = dict(coef=dict(save_coef=True, save_warp=False), saved_warp=dict(save_coef=False, save_warp=True),
                rewarp=dict(save_coef=False, save_warp=False))[bwd_path]
    K = G.t(c["K"])
    Ts = {f: G.t(c.T(f)).requires_grad_(f != "s") for f in c.sources_ids}
    P = torch.stack([G.F.compose_projection(K, Ts[f]) for f in c.sources_ids])
    srcs = [G.t(c.color(f)) for f in c.sources_ids]
    n = c.B * c.H * c.W
    ident = G.F.identity_loss(G.t(c.color(0)), srcs) if c.automask else None
    total = 0
    disps = []
    for s in range(c.n_scales):
        disp = G.t(c["disp_s%d" % s]).requires_grad_(True)
        disps.append(disp)
        noise = G.t(c["noise_s%d" % s]) if c.automask else None
        out = G.F.photometric_scale(disp, P, G.t(c.color(0)), srcs, G.t(c["inv_K"]), ident, noise,
                                    automask=c.automask, **keep)
        sm = G.F.smooth_loss(disp, G.t(c.color(0, s)))
        G.assert_close(sm, c["smooth_s%d" % s], "smooth s%d" % s)
        total = total + out["sum"][0] / n + 1e-3 * sm / (2 ** s)
    loss = total / c.n_scales
    loss.backward()
    G.assert_close(loss, c["loss"], "loss", rel=1e-5)
    for s in range(c.n_scales):
        G.assert_close(disps[s].grad, c["grad_disp_s%d" % s], "grad disp s%d" % s)
    for f in c.sources_ids:
        if f != "s":
            G.assert_close(Ts[f].grad, c["grad_T_%s" % f], "grad T %s" % f)


def test_fine_grained_ops_vs_golden(G):
    a = goldens.api()
    F = G.F
    for s in range(4):
        x = G.t(a["interp_in_s%d" % s]).requires_grad_(True)
        up = F.interpolate_bilinear(x, 24, 40)
        G.assert_bitexact(up, a["interp_out_s%d" % s], "interpolate s%d" % s)
        up.backward(G.t(a["interp_gout_s%d" % s]))
        G.assert_close(x.grad, a["interp_gin_s%d" % s], "interpolate bwd s%d" % s, rel=1e-5)
    for tag, (mn, mx) in {"train": (0.1, 100.0), "eval": (1e-3, 80)}.items():
        sd, dep = F.disparity2depth(G.t(a["d2d_in"]), mn, mx)
        G.assert_bitexact(sd, a["d2d_sd_" + tag], "scaled disp " + tag)
        G.assert_bitexact(dep, a["d2d_depth_" + tag], "depth " + tag)
    G.assert_bitexact(F.ssim(G.t(a["rl_pred"]), G.t(a["rl_targ"])), a["ssim_out"], "ssim")
    pred = G.t(a["rl_pred"]).requires_grad_(True)
    targ = G.t(a["rl_targ"]).requires_grad_(True)
    rl = F.reprojection_loss(pred, targ)
    G.assert_bitexact(rl, a["rl_out"], "reprojection loss")
    rl.backward(G.t(a["rl_gout"]))
    G.assert_close(pred.grad, a["rl_gpred"], "reprojection bwd pred")
    G.assert_close(targ.grad, a["rl_gtarg"], "reprojection bwd target")
    for s in range(4):
        d = G.t(a["sm_disp_s%d" % s]).requires_grad_(True)
        sm = F.smooth_loss(d, G.t(a["sm_color_s%d" % s]))
        G.assert_close(sm, a["sm_out_s%d" % s], "smooth s%d" % s)
        sm.backward()
        G.assert_close(d.grad, a["sm_gdisp_s%d" % s], "smooth grad s%d" % s)
    img = G.t(a["gs_img"]).requires_grad_(True)
    grid = G.t(a["gs_grid"]).requires_grad_(True)
    out = F.grid_sample_border(img, grid)
    G.assert_bitexact(out, a["gs_out"], "grid_sample")
    out.backward(G.t(a["gs_gout"]))
    G.assert_close(grid.grad, a["gs_ggrid"], "grid_sample bwd grid")
    G.assert_close(img.grad, a["gs_gimg"], "grid_sample bwd img")


def test_unfused_chain_vs_golden(G):
    """The reference's op-by-op pipeline (processor.py:141-162) through the fine-grained kernels."""
    F = G.F
    for name in goldens.FULL_CASES:
        c = goldens.Case(name)
        P = _P(G, c)
        for s in range(c.n_scales):
            up = F.interpolate_bilinear(G.t(c["disp_s%d" % s]), c.H, c.W)
            _, depth = F.disparity2depth(up, 0.1, 100.0)
            G.assert_bitexact(depth, c["depth_s%d" % s], "depth s%d" % s)
            cam = F.backproject(depth, G.t(c["inv_K"]))
            if s == 0:
                G.assert_bitexact(cam, c["cam_s0"], "cam")
            for i, f in enumerate(c.sources_ids):
                grid = F.project(cam, P[i], c.H, c.W)
                G.assert_bitexact(grid, c["grid_%s_s%d" % (f, s)], "grid %s s%d" % (f, s))
                warp = F.grid_sample_border(G.t(c.color(f)), grid)
                G.assert_bitexact(warp, c["warp_%s_s%d" % (f, s)], "warp %s s%d" % (f, s))


def _synth(B, H, W, S, seed):
    rng = np.random.RandomState(seed)
    colors = [(rng.randint(0, 256, size=(B, 3, H, W)).astype(np.float32) / np.float32(255.0)) for _ in range(S + 1)]
    # smooth the images a little so that windows are not pure noise
    K = np.array([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32)
    invK = np.linalg.pinv(K).astype(np.float32)
    K = np.repeat(K[None], B, 0)
    invK = np.repeat(invK[None], B, 0)
    Ts = []
    for f in range(S):
        T = np.repeat(np.eye(4, dtype=np.float32)[None], B, 0)
        T[:, :3, 3] = 0.05 * rng.randn(B, 3)
        T[:, :3, :3] += 0.01 * rng.randn(B, 3, 3).astype(np.float32)
        Ts.append(T.astype(np.float32))
    return colors, K, invK, Ts, rng


@pytest.mark.parametrize("B,H,W,S,scale", [(2, 192, 640, 2, 0), (2, 192, 640, 2, 2), (1, 192, 640, 3, 3),
                                           (1, 100, 150, 2, 0), (1, 320, 1024, 2, 1), (1, 192, 640, 4, 1),
                                           (3, 64, 68, 1, 0)])
def test_fused_vs_oracle_full_size(G, B, H, W, S, scale):
    """BASELINE-size tiles (192x640) and a ragged size, against the CPU oracle on seeded inputs."""
    from oracle import oracle as orc
    colors, K, invK, Ts, rng = _synth(B, H, W, S, seed=1234 + scale)
    h, w = (H >> scale, W >> scale) if H % 8 == 0 else (H, W)
    disp = rng.rand(B, 1, h, w).astype(np.float32)
    noise = rng.randn(B, S, H, W).astype(np.float32)
    P_ref = np.stack([orc.compose_projection(K, T) for T in Ts])
    ref = orc.photometric_fwd(disp, colors[0], colors[1:], invK, P_ref, noise, full=True)
    Kt = G.t(K)
    P = torch.stack([G.F.compose_projection(Kt, G.t(T)) for T in Ts])
    G.assert_bitexact(P, P_ref, "P")
    srcs = [G.t(x) for x in colors[1:]]
    ident = G.F.identity_loss(G.t(colors[0]), srcs)
    G.assert_bitexact(ident, ref["ident"], "ident")
    dt = G.t(disp).requires_grad_(True)
    Pt = P.detach().clone().requires_grad_(True)
    out = G.F.photometric_scale(dt, Pt, G.t(colors[0]), srcs, G.t(invK), ident, G.t(noise), need_to_opt=True,
                                need_depth=True, need_warp=True, need_reproj=True)
    G.assert_bitexact(out["depth"], ref["depth"], "depth")
    G.assert_bitexact(out["warp"], ref["warp"], "warp")
    G.assert_bitexact(out["reproj"], ref["reproj"], "reproj")
    G.assert_bitexact(out["to_opt"], ref["to_opt"], "to_opt")
    assert (out["idx"].cpu().numpy() == ref["idx"]).all(), "auto-mask indices"
    G.assert_close(out["sum"], np.array([ref["sum"]]), "sum", rel=1e-6)
    n = B * H * W
    (out["sum"][0] / n).backward()
    gd, gP = orc.photometric_bwd(disp, colors[0], colors[1:], invK, P_ref, ref["idx"], 1.0 / n)
    G.assert_close(dt.grad, gd, "grad disp")
    G.assert_close(Pt.grad, gP, "grad P")


def test_fused_without_automask_and_recompute_backward(G):
    """use_automasking=False (reprojection channels only, processor.py:197-198) and all three backward paths --
    coefficient maps (the default), saved warped colours, full re-warp -- against the oracle at 192x640."""
    from oracle import oracle as orc
    B, H, W, S = 1, 192, 640, 2
    colors, K, invK, Ts, rng = _synth(B, H, W, S, seed=99)
    disp = rng.rand(B, 1, H // 2, W // 2).astype(np.float32)
    P_ref = np.stack([orc.compose_projection(K, T) for T in Ts])
    ref = orc.photometric_fwd(disp, colors[0], colors[1:], invK, P_ref, None, automask=False, full=True)
    n = B * H * W
    gd, gP = orc.photometric_bwd(disp, colors[0], colors[1:], invK, P_ref, ref["idx"], 1.0 / n, automask=False)
    srcs = [G.t(x) for x in colors[1:]]
    for save_coef, save_warp in ((True, False), (False, True), (False, False)):
        tag = "save_coef=%s save_warp=%s" % (save_coef, save_warp)
        dt = G.t(disp).requires_grad_(True)
        Pt = G.t(P_ref).requires_grad_(True)
        out = G.F.photometric_scale(dt, Pt, G.t(colors[0]), srcs, G.t(invK), automask=False, need_to_opt=True,
                                    save_coef=save_coef, save_warp=save_warp)
        G.assert_bitexact(out["to_opt"], ref["to_opt"], "to_opt")
        assert (out["idx"].cpu().numpy() == ref["idx"]).all()
        (out["sum"][0] / n).backward()
        G.assert_close(dt.grad, gd, "grad disp (%s)" % tag)
        G.assert_close(Pt.grad, gP, "grad P (%s)" % tag)


def test_abi_reports_misuse_on_gpu(G):
    """Error behaviour through the binding: wrong dtype / non-contiguous / misaligned pointers raise, nothing crashes."""
    import ctypes as C
    from mdx import _lib
    x = torch.rand(1, 3, 32, 64, device=G.DEV)
    with pytest.raises(_lib.MdxError):
        _lib.ptr(x.double())
    with pytest.raises(_lib.MdxError):
        _lib.ptr(x[:, :, :, ::2])
    d = _lib.make_desc(1, 32, 64, 32, 64, 1, False, 0.1, 100.0)
    src = _lib.make_sources([x])
    buf = torch.rand(1 * 3 * 32 * 64 + 1, device=G.DEV)
    mis = buf[1:].view(1, 3, 32, 64)            # 4-byte aligned only
    out = torch.empty(1, 1, 32, 64, device=G.DEV)
    rc = _lib.lib().mdx_identity_loss(C.byref(d), C.c_void_p(mis.data_ptr()), C.byref(src), _lib.ptr(out), _lib.stream())
    assert rc == -6   # MDX_ERR_MISALIGNED



@pytest.mark.parametrize("saturate", [False, True])
def test_ssim_module_backward_vs_autograd(G, saturate):
    """model_loss.SSIM alone is differentiable (mdx_ssim_bwd): gradients wrt BOTH images for a random per-channel upstream
    against autograd of the reference formula (model_loss.py:28-41) evaluated in float64 -- no 0.85/3 factor, no L1 term.
    saturate: y = 1 - x over smooth fields drives (1 - SSIM)/2 towards the clamp's upper end over whole regions, and x = y
    regions sit on its lower end (value 0 up to rounding, gradient 0)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch_composite as tc
    from scipy.ndimage import gaussian_filter
    rng = np.random.RandomState(5 + int(saturate))
    B, Cc, H, W = 2, 3, 48, 80
    base = gaussian_filter(rng.randn(B, Cc, H, W), (0, 0, 3, 3))
    base = (base - base.min()) / (base.max() - base.min())
    x = (0.1 + 0.8 * base + 0.02 * rng.randn(B, Cc, H, W)).clip(0, 1).astype(np.float32)
    if saturate:
        y = (1.0 - x).astype(np.float32)
        y[:, :, :, : W // 3] = x[:, :, :, : W // 3]          # identical images: raw = 0, the closed lower end
    else:
        y = (0.1 + 0.8 * np.roll(base, 2, axis=3) + 0.02 * rng.randn(B, Cc, H, W)).clip(0, 1).astype(np.float32)
    up = rng.randn(B, Cc, H, W).astype(np.float32)
    xt, yt = G.t(x).requires_grad_(True), G.t(y).requires_grad_(True)
    out = G.F.ssim(xt, yt)
    out.backward(G.t(up))
    xr = torch.from_numpy(x).double().requires_grad_(True)
    yr = torch.from_numpy(y).double().requires_grad_(True)
    ref = tc.ssim(xr, yr)
    ref.backward(torch.from_numpy(up).double())
    # (the value is pinned bit for bit elsewhere; float32 against this float64 evaluation: the quotient's cancellation)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-3)
    if saturate:
        # (1 - SSIM) / 2 lies in [0, 1] mathematically: the clamp only ever acts at rounding level -- at its lower end where the
        # images are identical (SSIM = 1, a maximum: zero gradient), near its upper end where they are anti-correlated
        lo = (ref.detach()[..., 2:W // 3 - 2] <= 1e-12).float().mean()
        hi = ref.detach()[..., W // 3 + 2:].mean()
        assert float(lo) > 0.99 and float(hi) > 0.5, "the case reaches neither end of the clamp (%.3f, %.3f)" % (float(lo), float(hi))
        gmax = float(xt.grad.abs().max())
        assert float(xt.grad[..., 2:W // 3 - 2].abs().max()) <= 1e-3 * gmax, "gradient where the images are identical"
    # pixels whose float32 value sits within rounding of a clamp end may take the other branch than float64: compare away
    # from the ends' float32 neighbourhood by masking the (dilated) set of windows that are that close
    raw = ref.detach().numpy()
    near = (np.abs(raw - 1.0) < 1e-5) & (raw < 1.0) | ((raw > 0.0) & (raw < 1e-5))
    assert near.mean() < 0.01
    from scipy.ndimage import binary_dilation
    ok = ~binary_dilation(near, structure=np.ones((1, 1, 5, 5), bool))
    for name, g, r in (("x", xt.grad, xr.grad), ("y", yt.grad, yr.grad)):
        g, r = g.cpu().numpy().astype(np.float64), r.numpy()
        err = np.abs(g - r)[ok].max() / (np.abs(r).max() + 1e-30)
        assert err <= 1e-4, "d ssim / d %s: rel err %g" % (name, err)


@pytest.mark.parametrize("shape", [(2, 24, 40), (3, 17, 23), (1, 96, 320)])
def test_smoothness_two_launch_form_vs_float64_composite(G, shape):
    """csrc/smooth.hip (round 4: main pass on the disparity as given + finishing pass) against the reference formula
    (model_loss.py:77-88, 112-115) evaluated in float64: SmoothLoss, the bare EdgeAwareSmooth (normalize = 0: m = 1 exactly),
    a width that is not a multiple of 4 (the one-pixel-per-thread form), and a disparity with a NEGATIVE mean -- the identities
    |d_i/m - d_j/m| = |d_i - d_j| / |m| and sign(d_i/m - d_j/m) = sign(m) sign(d_i - d_j) hold for any m != 0."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch_composite as tc
    B, h, w = shape
    rng = np.random.RandomState(h * w)
    color = rng.rand(B, 3, h, w).astype(np.float32)
    for sign in (1.0, -1.0):
        disp = (sign * (0.05 + rng.rand(B, 1, h, w))).astype(np.float32)
        up = float(rng.rand() + 0.5)
        for normalize in (True, False):
            d = G.t(disp).requires_grad_(True)
            out = G.F.smooth_loss(d, G.t(color), normalize=normalize)
            (out * up).backward()
            dr = torch.from_numpy(disp).double().requires_grad_(True)
            cr = torch.from_numpy(color).double()
            if normalize:
                ref = tc.smooth_loss(dr, cr)
            else:
                gx = torch.abs(dr[:, :, :, :-1] - dr[:, :, :, 1:]) * torch.exp(-torch.abs(cr[:, :, :, :-1] - cr[:, :, :, 1:]).mean(1, True))
                gy = torch.abs(dr[:, :, :-1, :] - dr[:, :, 1:, :]) * torch.exp(-torch.abs(cr[:, :, :-1, :] - cr[:, :, 1:, :]).mean(1, True))
                ref = gx.mean() + gy.mean()
            (ref * up).backward()
            G.assert_close(out, ref.detach().numpy(), "smooth loss (sign %+d, normalize %d)" % (sign, normalize))
            G.assert_close(d.grad, dr.grad.numpy(), "smooth gradient (sign %+d, normalize %d)" % (sign, normalize))
