"""Geometry + sampling ops with the reference's names and signatures (model_layer/warp.py), backed by
hand-written gfx950 kernels through libmdx_hip.so.

    interpolate, grid_sample, upsample, disparity2depth, vector2translation, angle2rotation,
    param2matrix, Depth2PointCloud, PointCloud2Pixel

GPU tensors always take the kernels (and raise if libmdx_hip.so is missing -- there is no fallback for them).  CPU tensors --
the reference's own device pick on a machine without a GPU, model_train.py:28; BASELINE configs[0] -- take this package's
plain-PyTorch restatement of the same op sequences (mdx/composite.py): a dispatch on the tensor's device, never on whether
the library loaded.  Modes the training path never uses (non-bilinear interpolate, other grid_sample paddings) are forwarded
to torch.
"""
import torch
import torch.nn as nn
import torch.nn.functional as TF

from mdx import composite as C
from mdx import functional as F


def grid_sample(tensor, coords, padding_mode, align_corners):
    """reference: model_layer/warp.py:12-14 (bilinear)."""
    if padding_mode == "border" and align_corners:
        return F.grid_sample_border(tensor, coords) if tensor.is_cuda else C.grid_sample_border(tensor, coords)
    return TF.grid_sample(tensor, coords, padding_mode=padding_mode, align_corners=align_corners)


def interpolate(tensor, height, width, mode, align_corners):
    """reference: model_layer/warp.py:18-20."""
    if mode == "bilinear" and not align_corners and tensor.dim() == 4:
        return F.interpolate_bilinear(tensor, height, width) if tensor.is_cuda else C.interpolate_bilinear(tensor, height, width)
    if mode in ("nearest", "area", "nearest-exact"):
        return TF.interpolate(tensor, [height, width], mode=mode)
    return TF.interpolate(tensor, [height, width], mode=mode, align_corners=align_corners)


def upsample(tensor):
    """reference: model_layer/warp.py:24-25 (nearest x2, used by the decoder)."""
    return TF.interpolate(tensor, scale_factor=2, mode="nearest")


def disparity2depth(disparity, min_depth, max_depth):
    """reference: model_layer/warp.py:29-39 -> (scaled_disp, depth)."""
    if not disparity.is_cuda:
        return C.disparity2depth(disparity, min_depth, max_depth)
    return F.disparity2depth(disparity, min_depth, max_depth)


def vector2translation(translation_vector):
    """reference: model_layer/warp.py:43-61.  [N,1,3] -> [N,4,4]."""
    t = translation_vector.contiguous().view(-1, 3, 1)
    T = torch.eye(4, device=t.device, dtype=t.dtype).unsqueeze(0).repeat(t.shape[0], 1, 1)
    T[:, :3, 3:4] = t
    return T


def angle2rotation(anlge_axis):
    """reference: model_layer/warp.py:65-122 (Rodrigues).  [N,1,3] -> [N,4,4].  Same op sequence."""
    angle = torch.linalg.norm(anlge_axis, ord=2, dim=2, keepdim=True)
    axis = anlge_axis / (angle + 1e-5)
    cos, sin = torch.cos(angle), torch.sin(angle)
    Cc = 1 - cos
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xsin, ysin, zsin = x * sin, y * sin, z * sin
    xC, yC, zC = x * Cc, y * Cc, z * Cc
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    N = anlge_axis.shape[0]
    zero = torch.zeros(N, device=anlge_axis.device, dtype=anlge_axis.dtype)
    one = torch.ones_like(zero)

    def e(v):
        return v.reshape(N)
    rows = [
        torch.stack([e(x * xC + cos), e(xyC - zsin), e(zxC + ysin), zero], 1),
        torch.stack([e(xyC + zsin), e(y * yC + cos), e(yzC - xsin), zero], 1),
        torch.stack([e(zxC - ysin), e(yzC + xsin), e(z * zC + cos), zero], 1),
        torch.stack([zero, zero, zero, one], 1),
    ]
    return torch.stack(rows, 1)


def param2matrix(axisangle, translation, invert=False):
    """reference: model_layer/warp.py:126-153.  axisangle, translation [N,1,3] -> [N,4,4].
    float32 GPU tensors: one kernel each way (csrc/pose.hip) instead of ~40 element-wise ops; else the torch ops."""
    if axisangle.is_cuda and axisangle.dtype == torch.float32 and translation.dtype == torch.float32:
        return F.param2matrix(axisangle, translation, invert)
    R = angle2rotation(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = vector2translation(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


class Depth2PointCloud(nn.Module):
    """reference: model_layer/warp.py:193-246.  (depth [B,1,H,W], inv_K [B,4,4]) -> [B,4,H*W].
    The pixel grid is generated inside the kernel, so nothing is baked to `batch_size`."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inverse_intrinsic_matrix):
        depth = depth.reshape(-1, 1, self.height, self.width)
        if not depth.is_cuda:
            return C.backproject(depth, inverse_intrinsic_matrix)
        return F.backproject(depth, inverse_intrinsic_matrix)


class PointCloud2Pixel(nn.Module):
    """reference: model_layer/warp.py:250-269.  (cam [B,4,HW], K, T [B,4,4]) -> grid [B,H,W,2]."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, camera_coords, intrinsic_matrix, transformation_matrix):
        if not camera_coords.is_cuda:
            return C.project(camera_coords, intrinsic_matrix, transformation_matrix, self.height, self.width, self.eps)
        P = F.compose_projection(intrinsic_matrix, transformation_matrix)
        return F.project(camera_coords, P, self.height, self.width, self.eps)
