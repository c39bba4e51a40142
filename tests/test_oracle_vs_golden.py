"""CPU: the oracle (oracle/mdx_oracle.c) against golden vectors produced by the reference itself.

Per-pixel tensors and arg-min indices must be BIT-EXACT; scalars and gradients within 1e-4 rel
(the tolerance BASELINE.json's north_star states for float32 SSIM/smoothness).
"""
import numpy as np
import pytest

import goldens
from oracle import oracle as orc

REL = 1e-4


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bitexact(a, b, what):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    neq = bits(a) != bits(b)
    # +0 / -0 count as equal only if bits equal; report mismatches
    assert not neq.any(), "%s: %d / %d elements differ, max abs %g" % (
        what, neq.sum(), neq.size, np.abs(a - b).max())


def assert_close(a, b, what, rel=REL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max() / scale
    assert err <= rel, "%s: rel err %g > %g" % (what, err, rel)


@pytest.fixture(scope="module", params=goldens.CASES)
def case(request):
    return goldens.Case(request.param)


def _P(case):
    return np.stack([orc.compose_projection(case["K"], case.T(f)) for f in case.sources_ids])


def test_compose_projection_bitexact():
    for name in goldens.FULL_CASES:
        c = goldens.Case(name)
        for f in c.sources_ids:
            assert_bitexact(orc.compose_projection(c["K"], c.T(f)), c["P_%s" % f], "P_%s" % f)


def test_forward_per_scale(case):
    c = case
    P = _P(c)
    srcs = [c.color(f) for f in c.sources_ids]
    full = ("warp_%s_s0" % c.sources_ids[0]) in c
    for s in range(c.n_scales):
        noise = c["noise_s%d" % s] if c.automask else None
        out = orc.photometric_fwd(c["disp_s%d" % s], c.color(0), srcs, c["inv_K"], P, noise,
                                  automask=c.automask, full=True)
        assert_bitexact(out["depth"], c["depth_s%d" % s], "depth s%d" % s)
        if full:
            for i, f in enumerate(c.sources_ids):
                assert_bitexact(out["grid"][i], c["grid_%s_s%d" % (f, s)], "grid %s s%d" % (f, s))
                assert_bitexact(out["warp"][i], c["warp_%s_s%d" % (f, s)], "warp %s s%d" % (f, s))
            if "combined_s%d" % s in c:
                assert_bitexact(out["combined"], c["combined_s%d" % s], "combined s%d" % s)
        tgt = c["to_optimise_s%d" % s]
        assert_bitexact(out["to_opt"].reshape(tgt.shape), tgt, "to_optimise s%d" % s)
        if "idx_s%d" % s in c:
            assert (out["idx"] == c["idx_s%d" % s]).all(), "auto-mask indices s%d" % s


def test_loss_and_grads(case):
    c = case
    P = _P(c)
    srcs = [c.color(f) for f in c.sources_ids]
    total = 0.0
    n = c.B * c.H * c.W
    for s in range(c.n_scales):
        noise = c["noise_s%d" % s] if c.automask else None
        out = orc.photometric_fwd(c["disp_s%d" % s], c.color(0), srcs, c["inv_K"], P, noise,
                                  automask=c.automask)
        sm, gsm = orc.smooth_loss(c["disp_s%d" % s], c.color(0, s), need_grad=True)
        assert_close(sm, c["smooth_s%d" % s], "smooth s%d" % s)
        total += out["sum"] / n + 1e-3 * sm / (2 ** s)
        gd, gP = orc.photometric_bwd(c["disp_s%d" % s], c.color(0), srcs, c["inv_K"], P, out["idx"],
                                     1.0 / (c.n_scales * n), automask=c.automask)
        gd = gd + gsm * (1e-3 / (2 ** s) / c.n_scales)
        assert_close(gd, c["grad_disp_s%d" % s], "grad disp s%d" % s)
        if s == 0:
            gP_tot = gP.astype(np.float64)
        else:
            gP_tot += gP
    assert_close(total / c.n_scales, c["loss"], "loss", rel=1e-5)
    for i, f in enumerate(c.sources_ids):
        if f == "s":
            continue
        gT = orc.compose_projection_bwd(c["K"], gP_tot[i].astype(np.float32))
        assert_close(gT, c["grad_T_%s" % f], "grad T %s" % f)


def test_api_ops():
    a = goldens.api()
    for s in range(4):
        up = orc.upsample_bilinear(a["interp_in_s%d" % s], 24, 40)
        assert_bitexact(up, a["interp_out_s%d" % s], "interpolate s%d" % s)
        h, w = a["interp_in_s%d" % s].shape[2:]
        assert_close(orc.upsample_bilinear_bwd(a["interp_gout_s%d" % s], h, w), a["interp_gin_s%d" % s],
                     "interpolate bwd s%d" % s, rel=1e-6)
    for tag, (mn, mx) in {"train": (0.1, 100.0), "eval": (1e-3, 80)}.items():
        sd, dep = orc.disparity2depth(a["d2d_in"], mn, mx)
        assert_bitexact(sd, a["d2d_sd_" + tag], "scaled disp " + tag)
        assert_bitexact(dep, a["d2d_depth_" + tag], "depth " + tag)
    assert_bitexact(orc.ssim(a["rl_pred"], a["rl_targ"]), a["ssim_out"], "ssim")
    assert_bitexact(orc.reprojection_loss(a["rl_pred"], a["rl_targ"]), a["rl_out"], "reprojection loss")
    gp, gt = orc.reprojection_loss_bwd(a["rl_pred"], a["rl_targ"], a["rl_gout"], need_target=True)
    assert_close(gp, a["rl_gpred"], "reprojection bwd pred", rel=2e-5)
    assert_close(gt, a["rl_gtarg"], "reprojection bwd target", rel=2e-5)
    for s in range(4):
        v, g = orc.smooth_loss(a["sm_disp_s%d" % s], a["sm_color_s%d" % s], need_grad=True)
        assert_close(v, a["sm_out_s%d" % s], "smooth s%d" % s, rel=1e-5)
        assert_close(g, a["sm_gdisp_s%d" % s], "smooth grad s%d" % s, rel=1e-5)
    assert_bitexact(orc.grid_sample(a["gs_img"], a["gs_grid"]), a["gs_out"], "grid_sample")
    gg, gi = orc.grid_sample_bwd(a["gs_img"], a["gs_grid"], a["gs_gout"], need_img=True)
    assert_close(gg, a["gs_ggrid"], "grid_sample bwd grid", rel=1e-5)
    assert_close(gi, a["gs_gimg"], "grid_sample bwd img", rel=1e-5)
