"""TEST INFRASTRUCTURE: the loss path of one training step as plain PyTorch ops -- this build's own restatement of
compute.image2warping + compute.compute_loss (reference processor.py:139-218 with warp.py:12-39,193-269 and
model_loss.py:11-116), runnable on CPU tensors.  bench.py's cpu_baseline times it beside the oracle (SURVEY 8d asks for
both lines); tests/test_torch_composite.py checks it against the reference-made goldens.  Never imported by the product."""
import torch
import torch.nn.functional as F


def ssim(x, y):
    x, y = F.pad(x, (1, 1, 1, 1), mode="reflect"), F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = F.avg_pool2d(x, 3, 1), F.avg_pool2d(y, 3, 1)
    sig_x = F.avg_pool2d(x * x, 3, 1) - mu_x * mu_x
    sig_y = F.avg_pool2d(y * y, 3, 1) - mu_y * mu_y
    sig_xy = F.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + 0.01 ** 2) * (2 * sig_xy + 0.03 ** 2)
    d = (mu_x ** 2 + mu_y ** 2 + 0.01 ** 2) * (sig_x + sig_y + 0.03 ** 2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred, target):
    l1 = torch.abs(target - pred).mean(1, True)
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def smooth_loss(disp, color):
    disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
    gx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gx = gx * torch.exp(-torch.abs(color[:, :, :, :-1] - color[:, :, :, 1:]).mean(1, True))
    gy = gy * torch.exp(-torch.abs(color[:, :, :-1, :] - color[:, :, 1:, :]).mean(1, True))
    return gx.mean() + gy.mean()


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
