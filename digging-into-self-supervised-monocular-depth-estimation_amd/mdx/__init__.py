"""Core of the MI355X photometric path: ctypes binding of libmdx_hip.so + autograd functions."""
from ._lib import lib, MdxError, LIB_PATH  # noqa: F401
from . import functional  # noqa: F401
