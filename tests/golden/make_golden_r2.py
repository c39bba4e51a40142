#!/usr/bin/env python3
"""Round-2 golden vectors, recorded from the REFERENCE (runs only in the build container; /root/reference never travels).

    python tests/golden/make_golden_r2.py        # writes tests/golden/r2_*.npz  (data only: inputs + expected outputs)

  r2_full_192x640_b1.npz   compute.image2warping + compute.compute_loss (processor.py:139-218) at the BASELINE image size
                           (192x640, batch 1, frames [0,-1,1], scales 0 and 2), compact: uint8 colours, float16-exact
                           disparities and noise (the spy that captures torch.randn hands the reference the rounded
                           draw), uint8 arg-min indices, to_optimise of scale 0, loss, gradients.
  r2_decoders.npz          DepthDecoder (depth_decoder.py:54-112) and PoseDecoder (pose_decoder.py:13-58) with
                           closed-form parameters (tests/golden_params.py: the same function fills this build's modules,
                           so no 12 MB state dict is stored): inputs, outputs, input gradients, per-parameter gradient sums.
  r2_metrics.npz           compute_depth_error (numpy / torch) and compute_depth_metric (model_metric.py:19-105; loaded
                           behind a cv2 module object whose only use there is cv2.setNumThreads), and the reference's
                           point2depth (model_utility.py:128-197, np.int restored) on a synthetic calibration + scan
                           with duplicate pixels and the (row, 0) / (row-1, last) index collision of its sub2ind.
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import make_golden as G1   # noqa: E402  (loads the reference's warp / loss / processor behind stub packages)
from golden_params import fill_parameters   # noqa: E402
import fake_kitti   # noqa: E402

REF = G1.REF
WARP, LOSS, PROC = G1.WARP, G1.LOSS, G1.PROC


def f16_exact(t):
    return t.half().float()


def run_full_case(name="r2_full_192x640_b1", B=1, H=192, W=640, seed=21, scales=(0, 2), shifts=None, tx=None,
                  disp_noise=0.3, disp_lo=1.0, margin=16):
    """shifts / tx (round 3, tests/golden/make_golden_r3.py): source frame f is the target moved by shifts[f] pixels and
    its pose translates by tx[f] along x, so that the warp re-aligns most of it -- the reprojection channels then win
    on most pixels (few auto-masked).  Defaults reproduce the round-2 fixture byte for byte."""
    frame_ids = [0, -1, 1]
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed + 1000)
    opt = G1.Opt()
    opt.scales, opt.frame_ids, opt.height, opt.width = list(scales), frame_ids, H, W
    opt.min_depth, opt.max_depth, opt.disp_smoothness = 0.1, 100.0, 1e-3
    opt.use_automasking, opt.batch, opt.pose_type, opt.pose_frames = True, B, "separate", "pair"
    comp = PROC.compute(opt, torch.device("cpu"))
    setting = types.SimpleNamespace(inv_projection={0: WARP.Depth2PointCloud(B, H, W)},
                                    for_projection={0: WARP.PointCloud2Pixel(B, H, W)},
                                    loss={"reprojection": LOSS.ReprojectionLoss(), "edge_aware": LOSS.SmoothLoss()})
    store, inputs, outputs = {}, {}, {}
    # image-like colours: the target, and sources that are shifted copies of it plus a little texture (so that the
    # reprojection channels win on most pixels and the auto-mask takes the rest), uint8
    base = G1.synth_image_u8(gen, B, H, W + margin).float()
    for k, f in enumerate(frame_ids):
        sh = (shifts or {0: 8, -1: 5, 1: 11})[f]
        img = base[..., sh:sh + W] + (4.0 * torch.randn(B, 3, H, W, generator=gen) if f else 0)
        u8 = img.clamp(0, 255).round().to(torch.uint8)
        store["color_u8_%s" % f] = u8.numpy()
        inputs[("color", f, 0)] = u8.float() / 255.0
    for s in scales:
        if s:
            u8 = G1.synth_image_u8(gen, B, H >> s, W >> s)
            store["color0_u8_s%d" % s] = u8.numpy()
            inputs[("color", 0, s)] = u8.float() / 255.0
    K, invK = G1.make_K(B, H, W, "norm")
    inputs[("K", 0)], inputs[("inv_K", 0)] = K, invK
    store["K"], store["inv_K"] = K.numpy(), invK.numpy()
    disps = {}
    for s in scales:
        lo = torch.randn(B, 1, max(H >> (s + 3), 2), max(W >> (s + 3), 2), generator=gen)
        d = torch.sigmoid(disp_lo * torch.nn.functional.interpolate(lo, size=(H >> s, W >> s), mode="bilinear", align_corners=False)
                          + disp_noise * torch.randn(B, 1, H >> s, W >> s, generator=gen))
        d = f16_exact(d).requires_grad_(True)
        disps[s] = d
        outputs[("disp", s)] = d
        store["disp_f16_s%d" % s] = d.detach().half().numpy()
    Ts = {}
    for f in frame_ids[1:]:
        aa = 0.01 * torch.randn(B, 1, 3, generator=gen)
        tr = 0.03 * torch.randn(B, 1, 3, generator=gen)
        if tx is not None:
            aa, tr = 0.1 * aa, 0.1 * tr
            tr[..., 0] += tx[f]
        T = WARP.param2matrix(aa, tr, invert=(f < 0)).detach().clone().requires_grad_(True)
        Ts[f] = T
        outputs[("c2c", f, 0)] = T
        store["T_%s" % f] = T.detach().numpy()
    cap = {"noise": [], "min": []}
    real_randn, real_min = torch.randn, torch.min

    def randn_spy(*a, **k):
        r = f16_exact(real_randn(*a, **k))      # the reference draws N(0,1); it is handed the float16-exact value
        cap["noise"].append(r.clone())
        return r

    def min_spy(*a, **k):
        r = real_min(*a, **k)
        cap["min"].append((r[0].detach().clone(), r[1].detach().clone()))
        return r
    inputs, outputs = comp.image2warping(inputs, outputs, setting)
    torch.randn, torch.min = randn_spy, min_spy
    try:
        outputs = comp.compute_loss(inputs, outputs, setting)
    finally:
        torch.randn, torch.min = real_randn, real_min
    loss = outputs["loss"]
    loss.backward()
    store["loss"] = loss.detach().numpy()
    for k, s in enumerate(scales):
        store["noise_f16_s%d" % s] = cap["noise"][k].half().numpy()
        val, idx = cap["min"][k]
        store["idx_s%d" % s] = idx.numpy().astype(np.uint8)
        store["to_opt_sum_s%d" % s] = np.array(val.double().sum().item())
        if s == 0:
            store["to_optimise_s0"] = val.numpy()
            store["depth_rows_s0"] = outputs[("depth", 0, 0)].detach().numpy()[:, :, ::16]     # every 16th row
        store["grad_disp_s%d" % s] = disps[s].grad.numpy()
        store["smooth_s%d" % s] = setting.loss["edge_aware"](disp=disps[s].detach(), color=inputs[("color", 0, s)]).numpy()
    for f, T in Ts.items():
        store["grad_T_%s" % f] = T.grad.numpy()
    store["meta"] = np.array([B, H, W, 2, 1, len(scales)], dtype=np.int64)
    store["scales"] = np.array(scales, dtype=np.int64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **store)
    masked = float((cap["min"][0][1] < 2).float().mean())
    print("%-24s loss=%.8f masked %.1f%%  %7.1f KB" % (name, float(loss), 100 * masked, os.path.getsize(path) / 1024))


def run_decoders(name="r2_decoders"):
    dd = G1._load("_ref_depth_decoder", os.path.join(REF, "model_layer/depth_decoder.py"))
    pd = G1._load("_ref_pose_decoder", os.path.join(REF, "model_layer/pose_decoder.py"))
    st = {}
    gen = torch.Generator().manual_seed(5)
    num_ch_enc = np.array([64, 64, 128, 256, 512])
    B, H, W = 1, 64, 64
    dec = dd.DepthDecoder(num_ch_enc)
    fill_parameters(dec)
    feats = [(0.5 * torch.randn(B, int(c), H >> (k + 1), W >> (k + 1), generator=gen)).requires_grad_(True)
             for k, c in enumerate(num_ch_enc)]
    out = dec(feats)
    gouts = {s: torch.randn(out[("disp", s)].shape, generator=gen) for s in range(4)}
    total = sum((out[("disp", s)] * gouts[s]).sum() for s in range(4))
    total.backward()
    st["dec_keys"] = np.array(list(dec.state_dict().keys()))
    for k, f in enumerate(feats):
        st["dec_feat%d" % k], st["dec_gfeat%d" % k] = f.detach().numpy(), f.grad.numpy()
    for s in range(4):
        st["dec_disp%d" % s], st["dec_gout%d" % s] = out[("disp", s)].detach().numpy(), gouts[s].numpy()
    st["dec_gparam_sum"] = np.array([float(p.grad.double().sum()) for p in dec.parameters()])
    st["dec_gparam_abs"] = np.array([float(p.grad.double().abs().sum()) for p in dec.parameters()])
    # pose decoder: 1 input feature list, 2 frames to predict for (loader.py:85-86)
    pose = pd.PoseDecoder(num_ch_enc, 1, 2)
    fill_parameters(pose)
    feat = (0.5 * torch.randn(B, 512, 2, 3, generator=gen)).requires_grad_(True)
    aa, tr = pose([[feat]])
    ga, gt = torch.randn(aa.shape, generator=gen), torch.randn(tr.shape, generator=gen)
    ((aa * ga).sum() + (tr * gt).sum()).backward()
    st["pose_keys"] = np.array(list(pose.state_dict().keys()))
    st["pose_feat"], st["pose_gfeat"] = feat.detach().numpy(), feat.grad.numpy()
    st["pose_aa"], st["pose_tr"], st["pose_gaa"], st["pose_gtr"] = aa.detach().numpy(), tr.detach().numpy(), ga.numpy(), gt.numpy()
    st["pose_gparam_sum"] = np.array([float(p.grad.double().sum()) for p in pose.parameters()])
    st["pose_gparam_abs"] = np.array([float(p.grad.double().abs().sum()) for p in pose.parameters()])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **st)
    print("%-24s %d + %d parameters tensors  %7.1f KB" % (name, len(st["dec_keys"]), len(st["pose_keys"]),
                                                        os.path.getsize(path) / 1024))


def run_metrics(name="r2_metrics"):
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None           # model_metric.py:18 is the module's only use of cv2
    sys.modules["cv2"] = cv2
    metric = G1._load("_ref_metric", os.path.join(REF, "model_loss/model_metric.py"))
    st = {}
    rng = np.random.RandomState(3)
    gt = rng.uniform(1.0, 80.0, 5000).astype(np.float32)
    pred = (gt * np.exp(0.25 * rng.randn(5000))).astype(np.float32).clip(1e-3, 80)
    st["err_gt"], st["err_pred"] = gt, pred
    st["err_numpy"] = np.array(metric.compute_depth_error(gt, pred, "numpy"), dtype=np.float64)
    st["err_torch"] = np.array([float(v) for v in metric.compute_depth_error(torch.from_numpy(gt), torch.from_numpy(pred), "torch")])
    # compute_depth_metric: prediction [B,1,192,640] (float16-exact), sparse ground truth [B,1,375,1242]
    B = 2
    gen = torch.Generator().manual_seed(9)
    lo = torch.rand(B, 1, 12, 40, generator=gen) * 30 + 2
    pdepth = f16_exact(torch.nn.functional.interpolate(lo, size=(192, 640), mode="bilinear", align_corners=False))
    gtd = torch.zeros(B, 1, 375, 1242)
    n = 9000
    ys, xs = torch.randint(0, 375, (B, n), generator=gen), torch.randint(0, 1242, (B, n), generator=gen)
    vals = f16_exact(torch.rand(B, n, generator=gen) * 70 + 1.5)
    for b in range(B):
        # one value per pixel: a scatter with repeated indices keeps an unspecified one of them (the fixture could not
        # be re-derived bit for bit); the first draw of a pixel wins
        _, first = np.unique((ys[b] * 1242 + xs[b]).numpy(), return_index=True)
        first = torch.from_numpy(np.sort(first))
        gtd[b, 0, ys[b][first], xs[b][first]] = vals[b][first]
    out = metric.compute_depth_metric({("depth", 0): gtd.clone()}, {("depth", 0, 0): pdepth.clone()}, "torch")
    st["metric_pred_f16"] = pdepth.half().numpy()
    st["metric_gt_idx"] = torch.nonzero(gtd.reshape(-1)).reshape(-1).numpy().astype(np.int32)
    st["metric_gt_val_f16"] = gtd.reshape(-1)[torch.nonzero(gtd.reshape(-1)).reshape(-1)].half().numpy()
    st["metric_out"] = np.array([float(v) for v in out])
    # point2depth
    np.int = int                                  # removed from numpy >= 1.24; the reference calls .astype(np.int)
    try:
        util = G1._load("_ref_utility", os.path.join(REF, "model_utility.py"))
    except ImportError:
        sys.modules["scipy.misc"] = types.ModuleType("scipy.misc")      # imported, never used on this path
        import scipy
        scipy.misc = sys.modules["scipy.misc"]
        util = G1._load("_ref_utility", os.path.join(REF, "model_utility.py"))
    with tempfile.TemporaryDirectory() as root:
        fake_kitti.make(root, n_frames=1, seed=11)
        day = os.path.join(root, "2011_09_26")
        # the projection the reference forms (model_utility.py:150-153), to craft points on chosen pixels
        c2c = util.read_velo2cam(os.path.join(day, "calib_cam_to_cam.txt"))
        v2c = util.read_velo2cam(os.path.join(day, "calib_velo_to_cam.txt"))
        v2c = np.vstack((np.hstack((v2c["R"].reshape(3, 3), v2c["T"][..., np.newaxis])), np.array([0, 0, 0, 1.0])))
        R = np.eye(4)
        R[:3, :3] = c2c["R_rect_00"].reshape(3, 3)
        P = c2c["P_rect_02"].reshape(3, 4) @ R @ v2c

        def point_at(u, v, z):       # velodyne point that projects to raw image coordinates (u, v) at depth z
            return np.linalg.solve(P[:, :3], z * np.array([u, v, 1.0]) - P[:, 3])
        pts = [np.append(rng.uniform([4, -18, -1.8], [60, 18, 1.2]), 1.0) for _ in range(6000)]
        for (u, v) in ((300.2, 200.1), (900.4, 250.3), (45.0, 160.0)):       # several returns on one pixel
            for z in (31.0, 12.5, 47.0):
                pts.append(np.append(point_at(u, v, z), 1.0))
        n_cols = 1242
        for row in (180, 260):          # sub2ind's collision: (row, col 0) and (row - 1, last col) share an index
            pts.append(np.append(point_at(1.0, row + 1.0, 22.0), 1.0))
            pts.append(np.append(point_at(float(n_cols), row + 0.0, 9.0), 1.0))
            pts.append(np.append(point_at(float(n_cols), row + 0.0, 40.0), 1.0))
        pts = np.array(pts, dtype=np.float32)
        velo = os.path.join(root, "scan.bin")
        pts.tofile(velo)
        st["velo_points"] = pts
        st["calib_cam_to_cam"] = np.array(open(os.path.join(day, "calib_cam_to_cam.txt")).read())
        st["calib_velo_to_cam"] = np.array(open(os.path.join(day, "calib_velo_to_cam.txt")).read())
        for cam in (2, 3):
            for vd in (False, True):
                d = util.point2depth(day, velo, cam, vd)
                nz = np.flatnonzero(d)
                st["p2d_idx_c%d_v%d" % (cam, vd)] = nz.astype(np.int32)
                st["p2d_val_c%d_v%d" % (cam, vd)] = d.reshape(-1)[nz]
                st["p2d_shape"] = np.array(d.shape)
        # the per-image evaluation block of model_test.py:89-112 on one prediction: the steps are restated here (the
        # block is inline in the reference's inference(), not importable); the metric function is the reference's own
        # and the resize is torch's CPU bilinear kernel, which samples as cv2.resize's default does (cv2 is absent)
        gt_eval = util.point2depth(day, velo, 2, True).astype(np.float32)
        lo = torch.rand(1, 1, 12, 40, generator=gen) * 0.2 + 0.02
        pdisp = f16_exact(torch.nn.functional.interpolate(lo, size=(192, 640), mode="bilinear", align_corners=False))
        rs = torch.nn.functional.interpolate(pdisp, size=gt_eval.shape, mode="bilinear", align_corners=False)[0, 0].numpy()
        pdepth_e = 1 / rs
        mask = np.logical_and(gt_eval > 1e-3, gt_eval < 80.0)
        crop = np.zeros(mask.shape)
        crop[153:371, 44:1197] = 1
        mask = np.logical_and(mask, crop)
        pe, ge = pdepth_e[mask], gt_eval[mask]
        pe = pe * (np.median(ge) / np.median(pe))
        pe[pe < 1e-3] = 1e-3
        pe[pe > 80.0] = 80.0
        st["eval_disp_f16"] = pdisp[0, 0].half().numpy()
        st["eval_out"] = np.array(metric.compute_depth_error(ge, pe, "numpy"), dtype=np.float64)
        st["eval_resized_rows"] = rs[::25]
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **st)
    print("%-24s %7.1f KB  (point2depth non-zero pixels: %d)" % (name, os.path.getsize(path) / 1024, len(nz)))


if __name__ == "__main__":
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["full", "decoders", "metrics"]
    if "full" in which:
        run_full_case()
    if "decoders" in which:
        run_decoders()
    if "metrics" in which:
        run_metrics()
