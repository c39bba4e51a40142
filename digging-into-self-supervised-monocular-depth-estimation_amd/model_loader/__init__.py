"""KITTI datasets with the reference's class names (model_loader/__init__.py; wired at model_tool/loader.py:50-58)."""
from .kitti import KITTIDataset, KITTIMonoDataset_v2, KITTIMonoStereoDataset  # noqa: F401
