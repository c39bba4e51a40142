"""GPU test (MI355X) of what the step LEARNS -- the only proxy for BASELINE.json's abs_rel bar that exists without KITTI
data (reference: model_train.py:54-96 the loop, model_loss/model_metric.py:70-105 the metric, README.md:58,70 the bar).

The stand-in set of the throughput measurements has no geometry (source frames are shifted copies of the target, the ground
truth is random): per-step parity is pinned on it, but a sign or scale error in how the pieces are USED -- pose direction,
disparity range, which frame is warped onto which -- would pass every parity test.  Here the frames are rendered from rigid
textured scenes at known poses (model_tool/synthetic.py: scene; plain torch, nothing from oracle/), the product trainer runs
a few thousand Adam steps, and the train-time metric (median-scaled abs_rel against the scene's own depth) has to fall."""
import importlib

import numpy as np
import pytest
import torch

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")

pytestmark = pytest.mark.gpu

SCENES, BATCH, H, W = 64, 8, 96, 320


def _opt(graph=True, frame_ids=(0, -1, 1), **kw):
    bench = importlib.import_module("bench")
    opt = bench.make_opt(BATCH, height=H, width=W, frame_ids=frame_ids)
    opt.synthetic_geometry, opt.synthetic_length, opt.synthetic_pool = True, SCENES, SCENES
    opt.graph, opt.max_steps, opt.miopen_find = graph, 0, False
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


def _run(opt, epochs, seed=0):
    """-> per-epoch means of (loss, abs_rel, a1, auto-masked fraction at scale 0) over the fixed set of scenes."""
    from model_train import trainer
    torch.manual_seed(seed)
    tr = trainer(opt)
    tr.setting.set_train()
    batches = [{k: (v.to(tr.device) if torch.is_tensor(v) else v) for k, v in b.items()} for b in tr.setting.train_dataloader]
    assert len(batches) == SCENES // BATCH
    S = len(opt.frame_ids) - 1
    curves = []
    for _ in range(epochs):
        log = {k: [] for k in tr.control.metric_name}
        masked = []
        for b in batches:
            out = tr.train_step(dict(b))
            log = tr.control.metric(b, out, log)
            if ("automask", 0) in out:
                masked.append((out[("automask", 0)] < S).float().mean())
        mean = tr.control.epoch_means(log)
        curves.append((mean["loss"], mean["abs_rel"], mean["a1"], float(torch.stack(masked).mean()) if masked else float("nan")))
    return np.array(curves), tr


def test_scene_frames_are_consistent_with_their_depth_and_poses():
    """The generator's own contract, checked with the product's ops: warping a source frame with the scene's TRUE depth and pose
    (Depth2PointCloud / PointCloud2Pixel / grid_sample, reference model_layer/warp.py:193-269) reproduces the target far better
    than the unwarped source does -- so a network that finds depth and pose CAN bring the photometric loss down."""
    from model_layer import Depth2PointCloud, PointCloud2Pixel, grid_sample
    from model_tool.synthetic import SyntheticKITTI
    ds = SyntheticKITTI(4, [0, -1, 1, "s"], H, W, geometry=True)
    back, proj = Depth2PointCloud(1, H, W).cuda(), PointCloud2Pixel(1, H, W).cuda()
    for i in range(3):
        s = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in ds[i].items()}
        depth = s[("depth_dense", 0)][None]
        cam = back(depth, s[("inv_K", 0)][None])
        for f in (-1, 1, "s"):
            grid = proj(cam, s[("K", 0)][None], s[("pose_gt", f)][None])
            warped = grid_sample(s[("color", f, 0)][None], grid, "border", True)
            inside = (grid.abs() < 0.98).all(-1)[0]
            err = (warped[0] - s[("color", 0, 0)]).abs().mean(0)[inside].mean()
            ident = (s[("color", f, 0)] - s[("color", 0, 0)]).abs().mean(0)[inside].mean()
            assert float(err) < 0.035 and float(err) < 0.4 * float(ident), (i, f, float(err), float(ident))


def test_training_learns_depth_on_rigid_scenes():
    """A few thousand Adam steps of the product trainer (captured step, default layout plan, auto-masking on) on 64 rigid
    scenes: the median-scaled abs_rel of the train-time monitor falls below half its starting value and below a fixed bar,
    a1 rises, the loss falls, and the auto-mask lets go of pixels once the warp explains them better than 'nothing moved'."""
    curves, tr = _run(_opt(graph=True, learning_rate=2e-4), epochs=300)
    loss, abs_rel, a1, masked = curves.T
    first, last = curves[:3].mean(0), curves[-10:].mean(0)
    print("learning curve (epoch: loss abs_rel a1 masked):")
    for e in (0, 1, 2, 5, 10, 20, 50, 100, 150, 200, 250, 299):
        print("  %3d: %.4f %.4f %.4f %.4f" % ((e,) + tuple(curves[e])))
    assert np.isfinite(curves).all()
    assert last[0] < 0.75 * first[0], ("loss", first[0], last[0])
    assert last[1] < 0.5 * first[1] and last[1] < 0.25, ("abs_rel", first[1], last[1])
    assert last[2] > first[2] + 0.2, ("a1", first[2], last[2])
    assert last[3] < first[3], ("auto-masked fraction", first[3], last[3])


def test_fused_and_op_by_op_paths_follow_one_learning_curve():
    """The same seed through the fused kernels (the product default) and through the reference-shaped op-by-op path
    (mode "op_by_op": interpolate / disparity2depth / Depth2PointCloud / PointCloud2Pixel / grid_sample / ReprojectionLoss /
    SmoothLoss modules one by one, model_tool/processor.py) at a smaller size: one loss curve, to 1e-3 while the runs are in
    step and loosely (Adam amplifies rounding differences) afterwards."""
    global H, W, SCENES
    keep = (H, W, SCENES)
    try:
        H, W, SCENES = 64, 128, 16
        a, _ = _run(_opt(graph=False, use_automasking=False, fused=True), epochs=12)
        b, _ = _run(_opt(graph=False, use_automasking=False, fused=False, fused_train=False), epochs=12)
    finally:
        H, W, SCENES = keep
    np.testing.assert_allclose(a[:2, 0], b[:2, 0], rtol=1e-3)
    np.testing.assert_allclose(a[:, 0], b[:, 0], rtol=5e-2)
    assert a[-1, 0] < a[0, 0] and b[-1, 0] < b[0, 0]
