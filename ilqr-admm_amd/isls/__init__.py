"""isls -- batched DP-form iLQR-ADMM on AMD MI355X behind the reference's `isls` class surface.

Mirrors the module layout of the reference package (chenjianxing1/iLQR-ADMM `isls/__init__.py:1-4`:
`from .utils import *`, `SLS`, `iSLS`, `from .projections import *`) so that notebooks written against
the reference import the same names.  The numerics run in hand-written HIP kernels (csrc/) through the
C ABI of include/isls_hip.h; there is no CPU fallback.
"""
from . import _capi  # noqa: F401
