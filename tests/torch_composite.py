"""TEST INFRASTRUCTURE: the loss path of one training step as plain PyTorch ops -- this build's own restatement of
compute.image2warping + compute.compute_loss (reference processor.py:139-218 with warp.py:12-39,193-269 and
model_loss.py:11-116), runnable on CPU tensors, assembled from the package's own CPU op restatements (mdx/composite.py).
bench.py's cpu_baseline times it beside the oracle (SURVEY 8d asks for both lines); tests/test_torch_composite.py checks it
against the reference-made goldens.  Never imported by the product."""
import importlib

import torch
import torch.nn.functional as F

importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")


# the op-level restatements live in the package (mdx/composite.py: what model_layer / model_loss dispatch CPU tensors to)
from mdx.composite import reprojection_loss, smooth_loss, ssim  # noqa: E402,F401


def warp(disp, source, K, invK, T, H, W, min_depth=0.1, max_depth=100.0):
    B = disp.shape[0]
    disp = F.interpolate(disp, [H, W], mode="bilinear", align_corners=False)
    depth = 1 / (1 / max_depth + (1 / min_depth - 1 / max_depth) * disp)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W)], 0).unsqueeze(0).expand(B, 3, H * W)
    cam = depth.view(B, 1, -1) * torch.matmul(invK[:, :3, :3], pix)
    cam = torch.cat([cam, torch.ones(B, 1, H * W)], 1)
    q = torch.matmul(torch.matmul(K, T)[:, :3, :], cam)
    uv = (q[:, :2] / (q[:, 2:3] + 1e-7)).view(B, 2, H, W).permute(0, 2, 3, 1)
    grid = torch.stack([uv[..., 0] / (W - 1), uv[..., 1] / (H - 1)], -1)
    return F.grid_sample(source, (grid - 0.5) * 2, padding_mode="border", align_corners=True), depth


def loss_path(disps, colors, sources, K, invK, Ts, noises=None, automask=True, smoothness=1e-3):
    """disps: {scale: [B,1,h,w]}; colors: {scale: target colour pyramid}; sources, Ts: per source frame.
    Returns (loss, list of arg-min index maps)."""
    target = colors[0]
    B, _, H, W = target.shape
    total, idxs = 0, []
    for k, s in enumerate(sorted(disps)):
        reproj = torch.cat([reprojection_loss(warp(disps[s], src, K, invK, T, H, W)[0], target)
                            for src, T in zip(sources, Ts)], 1)
        if automask:
            ident = torch.cat([reprojection_loss(src, target) for src in sources], 1)
            noise = noises[k] if noises is not None else torch.randn(ident.shape)
            combined = torch.cat([ident + 0.00001 * noise, reproj], 1)
        else:
            combined = reproj
        if combined.shape[1] == 1:
            to_opt = combined
            idxs.append(None)
        else:
            to_opt, idx = torch.min(combined, 1)
            idxs.append(idx)
        total = total + to_opt.mean() + smoothness * smooth_loss(disps[s], colors[s]) / (2 ** s)
    return total / len(disps), idxs
