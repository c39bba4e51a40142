#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE hot path on CPU and records inputs + outputs.

Runs only in the build container (needs /root/reference; the reference never travels to the GPU
box).  It loads the reference's own modules by file path --
    model_layer/warp.py, model_loss/model_loss.py, model_tool/processor.py
-- behind stub packages that mirror the reference export lists (model_layer/__init__.py:6-11,
model_loss/__init__.py:1-2), then drives `compute.image2warping` + `compute.compute_loss`
(processor.py:139-218) on seeded synthetic inputs while capturing
  * every `torch.randn` noise tensor drawn at processor.py:195, and
  * the `(values, idxs)` pair returned by `torch.min` at processor.py:204,
without touching reference code.  Everything is stored as `.npz` data (inputs and expected outputs
only -- no reference source text).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("MDX_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    warp = _load("_ref_warp", os.path.join(REF, "model_layer/warp.py"))
    loss = _load("_ref_loss", os.path.join(REF, "model_loss/model_loss.py"))
    # stub packages so `from model_layer import *` etc. in processor.py resolve
    ml = types.ModuleType("model_layer")
    for k in ("interpolate", "grid_sample", "disparity2depth", "param2matrix",
              "Depth2PointCloud", "PointCloud2Pixel"):
        setattr(ml, k, getattr(warp, k))
    mlo = types.ModuleType("model_loss")
    for k in ("ReprojectionLoss", "SmoothLoss"):
        setattr(mlo, k, getattr(loss, k))
    sys.modules["model_layer"] = ml
    sys.modules["model_loss"] = mlo
    sys.modules["model_loader"] = types.ModuleType("model_loader")
    sys.modules["model_utility"] = types.ModuleType("model_utility")
    proc = _load("_ref_processor", os.path.join(REF, "model_tool/processor.py"))
    return warp, loss, proc


WARP, LOSS, PROC = load_reference()


# --------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------
def synth_image_u8(gen, B, H, W):
    """Blocky + gradient + texture image, uint8; has flat regions (sigma ~ 0 stresses SSIM's C2)."""
    bh, bw = max(H // 8, 1), max(W // 8, 1)
    coarse = torch.rand(B, 3, bh, bw, generator=gen)
    img = torch.nn.functional.interpolate(coarse, size=(H, W), mode="nearest")
    ramp = torch.linspace(0, 0.3, W).view(1, 1, 1, W) * torch.rand(B, 3, 1, 1, generator=gen)
    tex = 0.08 * torch.randn(B, 3, H, W, generator=gen)
    flat = (torch.rand(B, 1, bh, bw, generator=gen) < 0.35).float()
    flat = torch.nn.functional.interpolate(flat, size=(H, W), mode="nearest")
    img = img * 0.7 + ramp + tex * (1 - flat)
    return (img.clamp(0, 1) * 255).round().to(torch.uint8)


def make_K(B, H, W, variant="norm"):
    if variant == "norm":
        K = np.array([[0.58 * W, 0, 0.5 * W, 0], [0, 1.92 * H, 0.5 * H, 0],
                      [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    else:  # reference mono-loader behaviour (kitti_mono.py:326-327): row 1 scaled by width, floored
        K = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                     dtype=np.float32)
        K[0, :] *= W
        K[1, :] *= W
        K = np.floor(K).astype(np.float32)
        K[2, 2] = 1
        K[3, 3] = 1
    invK = np.linalg.pinv(K).astype(np.float32)
    K = torch.from_numpy(K).unsqueeze(0).repeat(B, 1, 1).contiguous()
    invK = torch.from_numpy(invK).unsqueeze(0).repeat(B, 1, 1).contiguous()
    return K, invK


class Opt:
    pass


def run_case(name, B, H, W, frame_ids, seed, pose_scale=(0.01, 0.01), automask=True,
             kvariant="norm", full=True, disp_mode="rand", n_scales=4):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed + 1000)
    opt = Opt()
    opt.scales = list(range(n_scales))
    opt.frame_ids = frame_ids
    opt.height, opt.width = H, W
    opt.min_depth, opt.max_depth = 0.1, 100.0
    opt.disp_smoothness = 1e-3
    opt.use_automasking = automask
    opt.batch = B
    opt.pose_type = "separate"
    opt.pose_frames = "pair"
    dev = torch.device("cpu")
    comp = PROC.compute(opt, dev)

    class Setting:
        pass
    setting = Setting()
    setting.inv_projection = {0: WARP.Depth2PointCloud(B, H, W)}
    setting.for_projection = {0: WARP.PointCloud2Pixel(B, H, W)}
    setting.loss = {"reprojection": LOSS.ReprojectionLoss(), "edge_aware": LOSS.SmoothLoss()}

    store = {}
    inputs, outputs = {}, {}
    # colours: full-res for every frame; pyramid for frame 0 (smoothness)
    for f in frame_ids:
        u8 = synth_image_u8(gen, B, H, W)
        store["color_u8_%s" % f] = u8.numpy()
        inputs[("color", f, 0)] = u8.float() / 255.0
    for s in opt.scales[1:]:
        u8 = synth_image_u8(gen, B, H >> s, W >> s)
        store["color0_u8_s%d" % s] = u8.numpy()
        inputs[("color", 0, s)] = u8.float() / 255.0
    K, invK = make_K(B, H, W, kvariant)
    inputs[("K", 0)], inputs[("inv_K", 0)] = K, invK
    store["K"], store["inv_K"] = K.numpy(), invK.numpy()

    disps = []
    for s in opt.scales:
        if disp_mode == "rand":
            d = torch.rand(B, 1, H >> s, W >> s, generator=gen)
        elif disp_mode == "sigmoid":
            d = torch.sigmoid(2.0 * torch.randn(B, 1, H >> s, W >> s, generator=gen))
        else:  # extreme: exact 0/1 plus random
            d = torch.rand(B, 1, H >> s, W >> s, generator=gen)
            m = torch.rand(B, 1, H >> s, W >> s, generator=gen)
            d = torch.where(m < 0.1, torch.zeros_like(d), d)
            d = torch.where(m > 0.9, torch.ones_like(d), d)
        d.requires_grad_(True)
        disps.append(d)
        outputs[("disp", s)] = d
        store["disp_s%d" % s] = d.detach().numpy()

    Ts = {}
    for f in frame_ids[1:]:
        if f == "s":
            T = torch.eye(4).unsqueeze(0).repeat(B, 1, 1)
            T[:, 0, 3] = 0.1 * (1 if seed % 2 else -1)
            inputs["stereo"] = T
            store["T_s"] = T.numpy()
        else:
            aa = pose_scale[0] * torch.randn(B, 1, 3, generator=gen)
            tr = pose_scale[1] * torch.randn(B, 1, 3, generator=gen)
            T = WARP.param2matrix(aa, tr, invert=(f < 0)).detach().clone()
            T.requires_grad_(True)
            Ts[f] = T
            outputs[("c2c", f, 0)] = T
            store["axisangle_%s" % f], store["translation_%s" % f] = aa.numpy(), tr.numpy()
            store["T_%s" % f] = T.detach().numpy()

    # ---- capture noise and the min ----
    cap = {"noise": [], "min": [], "cat": []}
    real_randn, real_min = torch.randn, torch.min

    def randn_spy(*a, **k):
        r = real_randn(*a, **k)
        cap["noise"].append(r.clone())
        return r

    def min_spy(*a, **k):
        r = real_min(*a, **k)
        cap["min"].append((a[0].detach().clone(), r[0].detach().clone(), r[1].detach().clone()))
        return r

    inputs, outputs = comp.image2warping(inputs, outputs, setting)
    torch.randn, torch.min = randn_spy, min_spy
    try:
        outputs = comp.compute_loss(inputs, outputs, setting)
    finally:
        torch.randn, torch.min = real_randn, real_min
    loss = outputs["loss"]
    loss.backward()

    S = len(frame_ids) - 1
    store["loss"] = loss.detach().numpy()
    for s in opt.scales:
        store["grad_disp_s%d" % s] = disps[s].grad.numpy()
        store["depth_s%d" % s] = outputs[("depth", 0, s)].detach().numpy()
        if automask:
            store["noise_s%d" % s] = cap["noise"][s].numpy()
        if len(cap["min"]) > s:
            comb, val, idx = cap["min"][s]
            store["to_optimise_s%d" % s] = val.numpy()
            assert idx.dtype == torch.int64
            store["idx_s%d" % s] = idx.numpy().astype(np.uint8)
            if full:
                store["combined_s%d" % s] = comb.numpy()
        # smoothness scalar, recomputed with the reference module (same call as processor.py:208)
        store["smooth_s%d" % s] = setting.loss["edge_aware"](
            disp=disps[s].detach(), color=inputs[("color", 0, s)]).numpy()
        if full:
            for f in frame_ids[1:]:
                store["warp_%s_s%d" % (f, s)] = outputs[("warp_color", f, s)].detach().numpy()
    if not len(cap["min"]):
        # single channel, no min (processor.py:201-202): record the map itself
        for s in opt.scales:
            pred = outputs[("warp_color", frame_ids[1], s)].detach()
            store["to_optimise_s%d" % s] = setting.loss["reprojection"](
                pred, inputs[("color", 0, 0)]).numpy()
    for f, T in Ts.items():
        store["grad_T_%s" % f] = T.grad.numpy()
    if full:
        # intermediate geometry for scale 0..: P = (K@T)[:, :3], camera points, grid
        for f in frame_ids[1:]:
            T = inputs["stereo"] if f == "s" else Ts[f].detach()
            store["P_%s" % f] = torch.matmul(K, T)[:, :3, :].numpy()
        for s in opt.scales:
            depth = outputs[("depth", 0, s)].detach()
            cam = setting.inv_projection[0](depth, invK)
            if s == 0:
                store["cam_s0"] = cam.detach().numpy()
            for f in frame_ids[1:]:
                T = inputs["stereo"] if f == "s" else Ts[f].detach()
                store["grid_%s_s%d" % (f, s)] = setting.for_projection[0](cam, K, T).detach().numpy()
    store["meta"] = np.array([B, H, W, S, int(automask), n_scales], dtype=np.int64)
    store["frame_ids"] = np.array([str(f) for f in frame_ids])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **store)
    print("%-28s loss=%.8f  %7.1f KB" % (name, float(loss), os.path.getsize(path) / 1024))


def run_api_case():
    """Golden vectors for the fine-grained API (warp.py:18-39,126-153; model_loss.py:92-116)."""
    gen = torch.Generator().manual_seed(77)
    st = {}
    # interpolate (processor.py:142) for each scale ratio, plus its gradient
    for s, (h, w) in enumerate([(24, 40), (12, 20), (6, 10), (3, 5)]):
        d = torch.rand(2, 1, h, w, generator=gen, requires_grad=True)
        up = WARP.interpolate(d, 24, 40, "bilinear", False)
        g = torch.randn(up.shape, generator=gen)
        up.backward(g)
        st["interp_in_s%d" % s], st["interp_out_s%d" % s] = d.detach().numpy(), up.detach().numpy()
        st["interp_gout_s%d" % s], st["interp_gin_s%d" % s] = g.numpy(), d.grad.numpy()
    # disparity2depth: training (0.1, 100) and eval (1e-3, 80; model_test.py:81)
    d = torch.rand(2, 1, 24, 40, generator=gen)
    d[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.5, 1e-6])
    st["d2d_in"] = d.numpy()
    for tag, (mn, mx) in {"train": (0.1, 100.0), "eval": (1e-3, 80)}.items():
        sd, dep = WARP.disparity2depth(d, mn, mx)
        st["d2d_sd_" + tag], st["d2d_depth_" + tag] = sd.numpy(), dep.numpy()
    # param2matrix (both invert flags) + grads
    aa = (0.05 * torch.randn(4, 1, 3, generator=gen)).requires_grad_(True)
    tr = (0.2 * torch.randn(4, 1, 3, generator=gen)).requires_grad_(True)
    st["p2m_aa"], st["p2m_tr"] = aa.detach().numpy(), tr.detach().numpy()
    for inv in (False, True):
        M = WARP.param2matrix(aa, tr, invert=inv)
        st["p2m_M_%d" % inv] = M.detach().numpy()
        gM = torch.randn(M.shape, generator=gen)
        ga, gt = torch.autograd.grad(M, (aa, tr), gM)
        st["p2m_gM_%d" % inv], st["p2m_gaa_%d" % inv], st["p2m_gtr_%d" % inv] = \
            gM.numpy(), ga.numpy(), gt.numpy()
    # ReprojectionLoss fwd + grad wrt prediction AND target (model_loss.py:97-103)
    pred = (synth_image_u8(gen, 2, 24, 40).float() / 255.0
            + 0.02 * torch.randn(2, 3, 24, 40, generator=gen)).clamp(0, 1).requires_grad_(True)
    targ = (synth_image_u8(gen, 2, 24, 40).float() / 255.0).requires_grad_(True)
    rl = LOSS.ReprojectionLoss()(pred, targ)
    g = torch.rand(rl.shape, generator=gen)
    rl.backward(g)
    st["rl_pred"], st["rl_targ"], st["rl_out"] = pred.detach().numpy(), targ.detach().numpy(), rl.detach().numpy()
    st["rl_gout"], st["rl_gpred"], st["rl_gtarg"] = g.numpy(), pred.grad.numpy(), targ.grad.numpy()
    st["ssim_out"] = LOSS.SSIM()(pred.detach(), targ.detach()).numpy()
    # SmoothLoss fwd + grad (model_loss.py:107-116)
    for s, (h, w) in enumerate([(24, 40), (12, 20), (6, 10), (3, 5)]):
        d = torch.rand(2, 1, h, w, generator=gen, requires_grad=True)
        c = synth_image_u8(gen, 2, h, w).float() / 255.0
        sm = LOSS.SmoothLoss()(disp=d, color=c)
        sm.backward()
        st["sm_disp_s%d" % s], st["sm_color_s%d" % s] = d.detach().numpy(), c.numpy()
        st["sm_out_s%d" % s], st["sm_gdisp_s%d" % s] = sm.detach().numpy(), d.grad.numpy()
    # grid_sample fwd + grad wrt grid and input (warp.py:12-14), incl. out-of-range coords
    img = torch.rand(2, 3, 24, 40, generator=gen, requires_grad=True)
    grid = (2.6 * torch.rand(2, 24, 40, 2, generator=gen) - 1.3).requires_grad_(True)
    out = WARP.grid_sample(img, grid, "border", True)
    g = torch.randn(out.shape, generator=gen)
    out.backward(g)
    st["gs_img"], st["gs_grid"], st["gs_out"] = img.detach().numpy(), grid.detach().numpy(), out.detach().numpy()
    st["gs_gout"], st["gs_ggrid"], st["gs_gimg"] = g.numpy(), grid.grad.numpy(), img.grad.numpy()
    path = os.path.join(OUT, "api_ops.npz")
    np.savez_compressed(path, **st)
    print("%-28s %7.1f KB" % ("api_ops", os.path.getsize(path) / 1024))


if __name__ == "__main__":
    torch.set_num_threads(1)
    run_case("mono_24x40_b2", 2, 24, 40, [0, -1, 1], seed=1)
    run_case("border_24x40_b2", 2, 24, 40, [0, -1, 1], seed=2, pose_scale=(0.15, 0.6),
             disp_mode="extreme")
    run_case("monobugK_24x40_b2", 2, 24, 40, [0, -1, 1], seed=3, kvariant="monobug",
             disp_mode="sigmoid")
    run_case("stereo_16x32_b1", 1, 16, 32, [0, -1, 1, "s"], seed=4, pose_scale=(0.02, 0.05))
    run_case("noautomask_16x32_b2", 2, 16, 32, [0, -1, 1], seed=5, automask=False)
    run_case("single_16x32_b2", 2, 16, 32, [0, 1], seed=6, automask=False)
    run_case("stereoonly_16x32_b2", 2, 16, 32, [0, "s"], seed=7)
    run_case("multi_64x160_b2", 2, 64, 160, [0, -1, 1], seed=8, pose_scale=(0.02, 0.08),
             full=False, disp_mode="sigmoid")
    run_api_case()
