"""Drop-in for the reference's `model_layer` package: same public names as reference model_layer/__init__.py:1-11
(four networks, six geometry / sampling ops)."""
from .depth_encoder import ResnetEncoder
from .depth_decoder import DepthDecoder
from .pose_decoder import PoseCNN, PoseDecoder
from .warp import (Depth2PointCloud, PointCloud2Pixel, disparity2depth, grid_sample, interpolate, param2matrix)

__all__ = ["ResnetEncoder", "DepthDecoder", "PoseCNN", "PoseDecoder", "interpolate", "grid_sample",
           "disparity2depth", "param2matrix", "Depth2PointCloud", "PointCloud2Pixel"]
