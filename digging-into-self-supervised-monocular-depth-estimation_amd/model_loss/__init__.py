"""Drop-in for the reference's model_loss package (export list: model_loss/__init__.py:1-3)."""
from .model_loss import ReprojectionLoss
from .model_loss import SmoothLoss
from .model_metric import *  # noqa: F401,F403
