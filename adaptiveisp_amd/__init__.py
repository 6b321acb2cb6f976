"""adaptiveisp_amd — MI355X-native (gfx950) AdaptiveISP hot path.

The per-pixel ISP filter stack runs as hand-written HIP kernels behind a C-ABI shared library
(csrc/libadaisp.so, header include/adaisp.h); this package is the host-side mirror of the
reference's Python surface for that path (Filter.process / Filter.forward / Agent.forward /
Value.forward, same names, arguments and state-dict keys). There is no CPU fallback: image ops
raise if the tensors are not on a HIP device or the library is missing.
"""
from . import _lib  # noqa: F401
from .config import cfg  # noqa: F401

__all__ = ["cfg", "_lib"]
__version__ = "0.1.0"
