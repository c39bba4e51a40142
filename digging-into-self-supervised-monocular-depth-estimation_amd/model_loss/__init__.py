"""Drop-in for the reference's `model_loss` package (reference model_loss/__init__.py:1-3): the two loss modules
and the depth metrics."""
from .model_loss import ReprojectionLoss, SmoothLoss
from .model_metric import compute_depth_error, compute_depth_metric

__all__ = ["ReprojectionLoss", "SmoothLoss", "compute_depth_error", "compute_depth_metric"]
