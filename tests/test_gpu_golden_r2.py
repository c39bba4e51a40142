"""GPU: round-2 pins against fixtures recorded from the reference (tests/golden/make_golden_r2.py):
  * the one-launch training kernel and the per-scale kernels at the BASELINE image size (192x640);
  * DepthDecoder through its hand-written glue path (csrc/glue.hip) and PoseDecoder against the reference modules;
  * the param2matrix kernel (csrc/pose.hip) against the reference's param2matrix values and gradients.
"""
import numpy as np
import pytest
import torch

import goldens
import goldens_r2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("name", goldens_r2.FULL_CASES)
@pytest.mark.parametrize("path", ["train_kernel", "per_scale"])
def test_full_size_vs_reference(G, path, name):
    c = goldens_r2.FullCase(name)
    K = G.t(c["K"])
    Ts = {f: G.t(c["T_%s" % f]).requires_grad_(f != "s") for f in c.sources_ids}      # inputs["stereo"] is a constant
    P = torch.stack([G.F.compose_projection(K, Ts[f]) for f in c.sources_ids])
    srcs = [G.t(c.color(f)) for f in c.sources_ids]
    tgt = G.t(c.color(0))
    n = c.B * c.H * c.W
    ident = G.F.identity_loss(tgt, srcs)
    disps = [G.t(c.disp(s)).requires_grad_(True) for s in c.scales]
    noises = [G.t(c.noise(s)) for s in c.scales]
    if path == "train_kernel":
        out = G.F.photometric_train(disps, P, tgt, srcs, G.t(c["inv_K"]), ident, noises, need_depth=True, need_to_opt=True)
        sums, idx, to0, depth = [out["sums"][k] for k in range(len(c.scales))], out["idx"], out["to_opt"][0], out["depth"]
    else:
        outs = [G.F.photometric_scale(disps[k], P, tgt, srcs, G.t(c["inv_K"]), ident, noises[k], need_to_opt=True,
                                      need_depth=True) for k in range(len(c.scales))]
        sums, idx, to0, depth = [o["sum"][0] for o in outs], [o["idx"] for o in outs], outs[0]["to_opt"], outs[0]["depth"]
    if "to_optimise_s0" in c:          # the fixtures that hold scale 0
        G.assert_bitexact(to0.reshape(c["to_optimise_s0"].shape), c["to_optimise_s0"], "to_optimise s0")
        G.assert_bitexact(depth[:, :, ::16], c["depth_rows_s0"], "depth rows s0")
    total = 0
    for k, s in enumerate(c.scales):
        assert (idx[k].cpu().numpy() == c["idx_s%d" % s]).all(), "auto-mask indices s%d" % s
        G.assert_close(sums[k].detach().cpu().numpy(), c["to_opt_sum_s%d" % s], "sum s%d" % s, rel=1e-6)
        sm = G.F.smooth_loss(disps[k], G.t(c.color(0, s)))
        G.assert_close(sm, c["smooth_s%d" % s], "smooth s%d" % s)
        total = total + sums[k] / n + 1e-3 * sm / (2 ** s)
    loss = total / len(c.scales)
    loss.backward()
    G.assert_close(loss, c["loss"], "loss", rel=1e-5)
    for k, s in enumerate(c.scales):
        G.assert_close(disps[k].grad, c["grad_disp_s%d" % s], "grad disp s%d" % s, elem=1e-4)
    for f in c.sources_ids:
        if f != "s":
            G.assert_close(Ts[f].grad, c["grad_T_%s" % f], "grad T %s" % f)


def test_depth_decoder_glue_path_vs_reference(G):
    from test_golden_r2_cpu import check_depth_decoder
    check_depth_decoder(G.DEV)


def test_pose_decoder_vs_reference_gpu(G):
    from test_golden_r2_cpu import check_pose_decoder
    check_pose_decoder(G.DEV)


def test_param2matrix_kernel_vs_reference(G):
    """mdx_param2matrix_{fwd,bwd} (one thread per pose, dual-number backward) against warp.py:126-153 and its autograd."""
    a = goldens.api()
    for inv in (False, True):
        aa = G.t(a["p2m_aa"]).requires_grad_(True)
        tr = G.t(a["p2m_tr"]).requires_grad_(True)
        M = G.F.param2matrix(aa, tr, invert=inv)
        G.assert_close(M, a["p2m_M_%d" % inv], "param2matrix invert=%d" % inv, rel=2e-6)
        M.backward(G.t(a["p2m_gM_%d" % inv]))
        G.assert_close(aa.grad, a["p2m_gaa_%d" % inv], "d axisangle invert=%d" % inv)
        G.assert_close(tr.grad, a["p2m_gtr_%d" % inv], "d translation invert=%d" % inv)


def test_depth_monitor_kernel_vs_reference(G):
    """csrc/monitor.hip (what compute_depth_metric runs on GPU tensors) against the reference's compute_depth_metric
    values (r2_metrics fixture), against the torch-op restatement on random data, and on an empty mask."""
    from test_golden_r2_cpu import metric_inputs
    from model_loss import compute_depth_metric
    z = goldens_r2.load("r2_metrics")
    inputs, outputs = metric_inputs(z, G.DEV)
    got = np.array([float(v) for v in compute_depth_metric(inputs, outputs, "torch")])
    G.assert_close(got, z["metric_out"], "depth monitor vs reference", rel=1e-5)
    # random dense-ish case: GPU kernel vs the torch-op form on the CPU
    g = torch.Generator().manual_seed(3)
    gt = torch.zeros(3, 1, 375, 1242)
    m = torch.rand(3, 1, 375, 1242, generator=g) < 0.2
    gt[m] = 0.5 + 90 * torch.rand(int(m.sum()), generator=g)
    pred = torch.rand(3, 1, 192, 640, generator=g) * 100 + 1e-4
    import model_loss.model_metric as mm
    keep, mm.METRIC_CAPACITY = mm.METRIC_CAPACITY, 1.0
    try:
        ref = np.array([float(v) for v in compute_depth_metric({("depth", 0): gt}, {("depth", 0, 0): pred}, "torch")])
    finally:
        mm.METRIC_CAPACITY = keep
    got = np.array([float(v) for v in compute_depth_metric({("depth", 0): gt.to(G.DEV)}, {("depth", 0, 0): pred.to(G.DEV)}, "torch")])
    G.assert_close(got, ref, "depth monitor vs torch ops", rel=2e-5)
    empty = compute_depth_metric({("depth", 0): torch.zeros(1, 1, 375, 1242, device=G.DEV)},
                                 {("depth", 0, 0): torch.ones(1, 1, 192, 640, device=G.DEV)}, "torch")
    assert all(np.isnan(float(v)) for v in empty)


def test_depth_monitor_dense_single_and_repeat(G):
    """the compacting monitor at the ends of its range: every pixel of the window valid (the compact arrays as large as the
    window, 23 slices per radix block), ONE valid pixel, values that all share their high bits -- against the torch-op form --
    and twice in a row on one workspace (the radix passes leave their histograms empty): bit-equal results."""
    from model_loss import compute_depth_metric
    import model_loss.model_metric as mm
    g = torch.Generator().manual_seed(11)
    cases = []
    dense = 1.0 + 60 * torch.rand(2, 1, 375, 1242, generator=g)
    cases.append(("dense", dense, torch.rand(2, 1, 192, 640, generator=g) * 50 + 0.5))
    one = torch.zeros(2, 1, 375, 1242)
    one[1, 0, 200, 700] = 17.5
    cases.append(("single", one, torch.rand(2, 1, 192, 640, generator=g) * 50 + 0.5))
    narrow = torch.zeros(2, 1, 375, 1242)
    m = torch.rand(2, 1, 375, 1242, generator=g) < 0.1
    narrow[m] = 10.0 + 1e-3 * torch.rand(int(m.sum()), generator=g)        # one exponent, a few mantissa patterns apart
    cases.append(("narrow", narrow, 10.0 + 1e-3 * torch.rand(2, 1, 192, 640, generator=g)))
    keep, mm.METRIC_CAPACITY = mm.METRIC_CAPACITY, 1.0
    try:
        for name, gt, pred in cases:
            ref = np.array([float(v) for v in compute_depth_metric({("depth", 0): gt}, {("depth", 0, 0): pred}, "torch")])
            a = torch.stack(list(compute_depth_metric({("depth", 0): gt.to(G.DEV)}, {("depth", 0, 0): pred.to(G.DEV)}, "torch")))
            b = torch.stack(list(compute_depth_metric({("depth", 0): gt.to(G.DEV)}, {("depth", 0, 0): pred.to(G.DEV)}, "torch")))
            assert torch.equal(a, b), name
            G.assert_close(a.cpu().numpy(), ref, "depth monitor, %s" % name, rel=2e-5)
    finally:
        mm.METRIC_CAPACITY = keep
