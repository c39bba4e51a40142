#!/usr/bin/env python3
"""Round-4 golden vector, recorded from the REFERENCE (runs only in the build container; /root/reference never travels).

    python tests/golden/make_golden_r4.py        # writes tests/golden/r4_full_192x640_b1_stereo.npz (data only)

  r4_full_192x640_b1_stereo.npz   compute.image2warping + compute.compute_loss (processor.py:139-218) at the BASELINE image
                                  size with frame_ids [0, -1, 1, "s"] -- BASELINE configs[4], three source frames, the stereo
                                  one through inputs["stereo"] (processor.py:148-149) -- for scales 0 and 3.  This is the
                                  shape the training kernel's LOW form (S >= 3) runs: the fixture pins it to the reference
                                  itself, not only to the oracle.  Same compact encoding as rounds 2-3 (uint8 colours,
                                  float16-exact disparities and noise, uint8 indices, to_optimise of scale 0, loss, gradients).
"""
import os
import types

import numpy as np
import torch

import make_golden_r2 as R2

G1, WARP, LOSS, PROC, f16_exact, HERE = R2.G1, R2.WARP, R2.LOSS, R2.PROC, R2.f16_exact, R2.HERE


def run_stereo_case(name="r4_full_192x640_b1_stereo", B=1, H=192, W=640, seed=44, scales=(0, 3), margin=48):
    frame_ids = [0, -1, 1, "s"]
    shifts = {0: 24, -1: 12, 1: 36, "s": 4}          # each source is the target moved by that many pixels (+ texture)
    tx = {-1: -0.0065, 1: -0.0065}                     # ... and its pose translates so that the warp re-aligns most of it
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
    base = G1.synth_image_u8(gen, B, H, W + margin).float()
    for f in frame_ids:
        sh = shifts[f]
        img = base[..., sh:sh + W] + (4.0 * torch.randn(B, 3, H, W, generator=gen) if f != 0 else 0)
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
        d = torch.sigmoid(0.35 * torch.nn.functional.interpolate(lo, size=(H >> s, W >> s), mode="bilinear", align_corners=False)
                          + 0.05 * torch.randn(B, 1, H >> s, W >> s, generator=gen))
        d = f16_exact(d).requires_grad_(True)
        disps[s] = d
        outputs[("disp", s)] = d
        store["disp_f16_s%d" % s] = d.detach().half().numpy()
    Ts = {}
    for f in frame_ids[1:]:
        if f == "s":
            # kitti_stereo.py:249-256: the identity with [0, 3] = +-0.1 (the baseline); a constant of the data layer
            T = torch.eye(4).repeat(B, 1, 1)
            T[:, 0, 3] = -0.1
            inputs["stereo"] = T
            store["T_s"] = T.numpy()
            continue
        aa = 0.001 * torch.randn(B, 1, 3, generator=gen)
        tr = 0.003 * torch.randn(B, 1, 3, generator=gen)
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
    S = len(frame_ids) - 1
    for k, s in enumerate(scales):
        assert tuple(cap["noise"][k].shape) == (B, S, H, W)
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
    store["meta"] = np.array([B, H, W, S, 1, len(scales)], dtype=np.int64)
    store["scales"] = np.array(scales, dtype=np.int64)
    store["sources"] = np.array([str(f) for f in frame_ids[1:]])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **store)
    idx0 = cap["min"][0][1]
    print("%-30s loss=%.8f masked %.1f%%  arg-min shares %s  %7.1f KB" % (
        name, float(loss), 100 * float((idx0 < S).float().mean()),
        [round(float((idx0 == S + j).float().mean()), 3) for j in range(S)], os.path.getsize(path) / 1024))


if __name__ == "__main__":
    torch.set_num_threads(4)
    run_stereo_case()
