"""MI355X-native photometric training path (drop-in for the reference's model_layer / model_loss /
model_tool API).

Importing this package puts its own directory on sys.path so that the drop-in top-level packages
it contains -- `model_layer`, `model_loss`, `model_tool` (same names and export lists as the
reference: model_layer/__init__.py:1-11, model_loss/__init__.py:1-3, model_tool/__init__.py:1-3)
-- and the core `mdx` package resolve by their reference names:

    import importlib; importlib.import_module("digging-into-self-supervised-monocular-depth-estimation_amd")
    from model_layer import *      # exactly what the reference's model_train.py:18 does
"""
import os
import sys

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
if PACKAGE_DIR not in sys.path:
    sys.path.insert(0, PACKAGE_DIR)

__version__ = "0.1.0"


def build(force=False, verbose=False):
    """Compile libmdx_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    from build import build as _build  # noqa: resolved through PACKAGE_DIR
    return _build(force=force, verbose=verbose)


def install_miopen_db(rank=None):
    """See mdx/tuning.py: tuned MIOpen find-db for the networks of this path (gfx950)."""
    from mdx.tuning import install_miopen_db as _install
    return _install(rank)
