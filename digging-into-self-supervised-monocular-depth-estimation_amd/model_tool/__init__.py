"""Drop-in for the reference's `model_tool` package (reference model_tool/__init__.py:1-3)."""
from .loader import setting
from .logger import control
from .processor import compute

__all__ = ["setting", "control", "compute"]
