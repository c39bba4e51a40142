"""Drop-in for the reference's `model_tool` package: exposes `setting` (loaders, networks, losses, optimiser, DDP),
`control` (metrics, printing, checkpoints) and `compute` (the step driver) -- the three names the reference's
trainer star-imports (reference model_tool/__init__.py:1-3, model_train.py:20)."""
from . import loader as _loader, logger as _logger, processor as _processor

setting, control, compute = _loader.setting, _logger.control, _processor.compute
__all__ = ["setting", "control", "compute"]
