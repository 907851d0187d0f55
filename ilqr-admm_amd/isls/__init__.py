"""isls -- batched DP-form iLQR-ADMM on AMD MI355X behind the reference's `isls` class surface.

Mirrors the public names of the reference package (chenjianxing1/iLQR-ADMM `isls/__init__.py:1-4`:
`from .utils import *`, `SLS`, `iSLS`, `from .projections import *`) so that code written against the reference
imports the same names.  The numerics run in hand-written HIP kernels (csrc/) through the C ABI of
include/isls_hip.h; there is no CPU fallback: constructing a solver without the built library or without a
HIP device raises.
"""
from . import _capi, costs, models  # noqa: F401
from .admm import ADMM  # noqa: F401
from .utils import get_double_integrator_AB, find_mus, find_precs  # noqa: F401
from .projections import *  # noqa: F401,F403
from .projections import Box, identify_box  # noqa: F401


def __getattr__(name):
    # `iSLS` / `SLS` pull in torch and the HIP library: import them on first use so that the pure-numpy parts
    # (projections, utils, _capi) stay importable on machines without a GPU runtime (tests -m "not gpu")
    if name in ("iSLS", "SLS", "Engine"):
        from . import engine, isls as _isls, sls as _sls
        return {"iSLS": _isls.iSLS, "SLS": _sls.SLS, "Engine": engine.Engine}[name]
    raise AttributeError(name)
