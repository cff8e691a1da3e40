"""MI355X-native SqueezeDet hot path (inference + training) behind the reference's
``nn.Module`` / ``Detector`` / ``Trainer`` surface.  See DESIGN.md.

Importing this package never touches the GPU (fork-safe); the HIP library
``csrc/libsqdhip.so`` is loaded on first use and its absence is a hard error -- there is no
CPU fallback in the product path.
"""
from . import boxes, config, synthetic  # noqa: F401
from .config import make_cfg  # noqa: F401

__version__ = "0.1.0"
