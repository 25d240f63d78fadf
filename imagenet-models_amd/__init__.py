"""MI355X-native engine for the GA-ConvNeXt / GA / MAP ImageNet training hot path.

Python host code (timm-style registry, model containers, train/validate glue) over hand-written HIP kernels for
gfx950 reached through the C ABI in include/gaext.h.  There is no CPU or eager-PyTorch fallback: compute entry
points raise if csrc/libgaext.so is missing.
"""
from . import _lib  # noqa: F401
from . import ops  # noqa: F401

__version__ = '0.1.0'
