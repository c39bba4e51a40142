"""Loss modules with the reference's names and call signatures (model_loss/model_loss.py:11-116),
backed by gfx950 kernels through libmdx_hip.so.  GPU float32 only -- no CPU fallback."""
import torch.nn as nn

from mdx import functional as F
from mdx._lib import MdxError


def _require_gpu(t, what):
    if not t.is_cuda:
        raise MdxError("%s: expected a CUDA/HIP tensor, got %s (this build has no CPU fallback)" % (what, t.device))


class SSIM(nn.Module):
    """reference: model_loss/model_loss.py:11-41.  3x3 average-pool SSIM with reflection padding,
    clamp((1-SSIM)/2, 0, 1).  Differentiable in both images, like the reference's module (closed-form backward,
    mdx_ssim_bwd)."""

    def __init__(self):
        super().__init__()
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, image1, image2):
        _require_gpu(image1, "SSIM")
        return F.ssim(image1, image2)


class EdgeAwareSmooth(nn.Module):
    """reference: model_loss/model_loss.py:45-88 (no mean-normalisation)."""

    def forward(self, disparity, image):
        _require_gpu(disparity, "EdgeAwareSmooth")
        return F.smooth_loss(disparity, image, normalize=False)


class ReprojectionLoss(nn.Module):
    """reference: model_loss/model_loss.py:92-103.  0.85*mean_c(SSIM) + 0.15*mean_c|target-pred| -> [B,1,H,W]."""

    def __init__(self):
        super().__init__()
        self.ssim = SSIM()

    def forward(self, prediction, target):
        _require_gpu(prediction, "ReprojectionLoss")
        return F.reprojection_loss(prediction, target)


class SmoothLoss(nn.Module):
    """reference: model_loss/model_loss.py:107-116.  Called as loss(disp=..., color=...) (processor.py:208)."""

    def __init__(self):
        super().__init__()
        self.edge_aware_smooth = EdgeAwareSmooth()

    def forward(self, disp, color):
        _require_gpu(disp, "SmoothLoss")
        return F.smooth_loss(disp, color, normalize=True)
