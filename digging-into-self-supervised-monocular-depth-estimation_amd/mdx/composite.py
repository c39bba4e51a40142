"""The hot-path ops as plain PyTorch op sequences, for CPU tensors only.

The reference picks its device at start-up (`'cuda' if torch.cuda.is_available() else 'cpu'`, model_train.py:28) and BASELINE
configs[0] is its CPU-runnable case.  model_layer / model_loss hand CPU tensors to the functions below -- this package's own
restatement of the reference's op sequences (warp.py:12-39,193-269; model_loss.py:11-116), checked against the reference-made
goldens by tests/test_torch_composite.py -- so that the same scripts run on a machine without a GPU.

This is a dispatch on the TENSOR'S DEVICE, not a fallback: a CUDA/HIP tensor always goes to libmdx_hip.so and raises if the
library is missing (mdx/_lib.py); nothing here is ever reached with a GPU tensor, and nothing under oracle/ is ever imported.
"""
import warnings

import torch
import torch.nn.functional as TF

_told = False


def _notice():
    global _told
    if not _told:
        _told = True
        warnings.warn("mdx: CPU tensors -- running the plain-PyTorch composite of the hot path (the reference's own CPU path, "
                      "model_train.py:28); the gfx950 kernels are not involved", stacklevel=3)


def interpolate_bilinear(x, H, W):
    _notice()
    return TF.interpolate(x, [H, W], mode="bilinear", align_corners=False)


def disparity2depth(disp, min_depth, max_depth):
    """reference warp.py:29-39 -> (scaled_disp, depth)."""
    _notice()
    lo, hi = 1 / max_depth, 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled


def backproject(depth, invK):
    """reference warp.py:193-246: depth [B,1,H,W], inv_K [B,4,4] -> [B,4,H*W]."""
    _notice()
    B, _, H, W = depth.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype), indexing="ij")
    pix = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(H * W, dtype=depth.dtype)], 0).unsqueeze(0).expand(B, 3, H * W)
    cam = depth.view(B, 1, -1) * torch.matmul(invK[:, :3, :3], pix)
    return torch.cat([cam, torch.ones(B, 1, H * W, dtype=depth.dtype)], 1)


def project(cam, K, T, H, W, eps=1e-7):
    """reference warp.py:250-269: cam [B,4,HW], K, T [B,4,4] -> grid [B,H,W,2] in [-1, 1]."""
    _notice()
    B = cam.shape[0]
    q = torch.matmul(torch.matmul(K, T)[:, :3, :], cam)
    uv = (q[:, :2] / (q[:, 2:3] + eps)).view(B, 2, H, W).permute(0, 2, 3, 1)
    grid = torch.stack([uv[..., 0] / (W - 1), uv[..., 1] / (H - 1)], -1)
    return (grid - 0.5) * 2


def grid_sample_border(img, grid):
    _notice()
    return TF.grid_sample(img, grid, padding_mode="border", align_corners=True)


def ssim(x, y):
    """reference model_loss.py:11-41."""
    _notice()
    x, y = TF.pad(x, (1, 1, 1, 1), mode="reflect"), TF.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x, mu_y = TF.avg_pool2d(x, 3, 1), TF.avg_pool2d(y, 3, 1)
    sig_x = TF.avg_pool2d(x * x, 3, 1) - mu_x * mu_x
    sig_y = TF.avg_pool2d(y * y, 3, 1) - mu_y * mu_y
    sig_xy = TF.avg_pool2d(x * y, 3, 1) - mu_x * mu_y
    n = (2 * mu_x * mu_y + 0.01 ** 2) * (2 * sig_xy + 0.03 ** 2)
    d = (mu_x ** 2 + mu_y ** 2 + 0.01 ** 2) * (sig_x + sig_y + 0.03 ** 2)
    return torch.clamp((1 - n / d) / 2, 0, 1)


def reprojection_loss(pred, target):
    """reference model_loss.py:92-103."""
    l1 = torch.abs(target - pred).mean(1, True)
    return 0.85 * ssim(pred, target).mean(1, True) + 0.15 * l1


def smooth_loss(disp, color, normalize=True):
    """reference model_loss.py:45-88 (normalize=False) / 107-116 (normalize=True)."""
    _notice()
    if normalize:
        disp = disp / (disp.mean(2, True).mean(3, True) + 1e-7)
    gx = torch.abs(disp[:, :, :, :-1] - disp[:, :, :, 1:])
    gy = torch.abs(disp[:, :, :-1, :] - disp[:, :, 1:, :])
    gx = gx * torch.exp(-torch.abs(color[:, :, :, :-1] - color[:, :, :, 1:]).mean(1, True))
    gy = gy * torch.exp(-torch.abs(color[:, :, :-1, :] - color[:, :, 1:, :]).mean(1, True))
    return gx.mean() + gy.mean()
