"""Drop-in for the reference's model_layer package (same export list: model_layer/__init__.py:1-11)."""
from .depth_encoder import ResnetEncoder
from .depth_decoder import DepthDecoder
from .pose_decoder import PoseCNN
from .pose_decoder import PoseDecoder

from .warp import interpolate
from .warp import grid_sample
from .warp import disparity2depth
from .warp import param2matrix
from .warp import Depth2PointCloud
from .warp import PointCloud2Pixel
