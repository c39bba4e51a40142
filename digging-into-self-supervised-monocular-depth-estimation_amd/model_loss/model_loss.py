"""Loss modules with the reference's names and call signatures (model_loss/model_loss.py:11-116),
backed by gfx950 kernels through libmdx_hip.so for GPU tensors (no fallback for them: a missing library raises); CPU
tensors -- the reference's device pick on a machine without a GPU, BASELINE configs[0] -- take the package's plain-PyTorch
restatement of the same op sequences (mdx/composite.py)."""
import torch.nn as nn

from mdx import composite as C
from mdx import functional as F


class SSIM(nn.Module):
    """reference: model_loss/model_loss.py:11-41.  3x3 average-pool SSIM with reflection padding,
    clamp((1-SSIM)/2, 0, 1).  Differentiable in both images, like the reference's module (closed-form backward,
    mdx_ssim_bwd)."""

    def __init__(self):
        super().__init__()
        self.C1 = 0.01 ** 2
        self.C2 = 0.03 ** 2

    def forward(self, image1, image2):
        return F.ssim(image1, image2) if image1.is_cuda else C.ssim(image1, image2)


class EdgeAwareSmooth(nn.Module):
    """reference: model_loss/model_loss.py:45-88 (no mean-normalisation)."""

    def forward(self, disparity, image):
        if not disparity.is_cuda:
            return C.smooth_loss(disparity, image, normalize=False)
        return F.smooth_loss(disparity, image, normalize=False)


class ReprojectionLoss(nn.Module):
    """reference: model_loss/model_loss.py:92-103.  0.85*mean_c(SSIM) + 0.15*mean_c|target-pred| -> [B,1,H,W]."""

    def __init__(self):
        super().__init__()
        self.ssim = SSIM()

    def forward(self, prediction, target):
        if not prediction.is_cuda:
            return C.reprojection_loss(prediction, target)
        return F.reprojection_loss(prediction, target)


class SmoothLoss(nn.Module):
    """reference: model_loss/model_loss.py:107-116.  Called as loss(disp=..., color=...) (processor.py:208)."""

    def __init__(self):
        super().__init__()
        self.edge_aware_smooth = EdgeAwareSmooth()

    def forward(self, disp, color):
        if not disp.is_cuda:
            return C.smooth_loss(disp, color, normalize=True)
        return F.smooth_loss(disp, color, normalize=True)
