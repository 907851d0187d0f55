"""ctypes binding of the C ABI declared in include/isls_hip.h.

`Kernels` marshals numpy arrays or torch tensors into the argument structs of the library.  The
product instantiates it on `csrc/libisls_hip.so` (device pointers + a HIP stream); the tests
instantiate the very same class on the CPU oracle (`oracle/liboracle_isls.so`, host pointers, no
stream) so that both sides are called with identical marshaling.

There is NO fallback: if the HIP library is missing, `load_hip_library()` raises.
"""
import ctypes as C
import os

import numpy as np

try:  # torch is plumbing (device memory, streams); the binding itself works on numpy too
    import torch
except Exception:  # pragma: no cover
    torch = None

ABI_VERSION = 107            # ISLS_VERSION of include/isls_hip.h these ctypes structs mirror
OK, ERR_ARG, ERR_UNSUPPORTED, ERR_LAUNCH = 0, -1, -2, -3
ST_NOT_PD, ST_NAN_COST, ST_LS_REJECT = 1, 2, 4
SOLVE_CHOL, SOLVE_INV = 0, 1
MODEL_LTI, MODEL_ARM3R, MODEL_CAR, MODEL_DI, MODEL_TASSA = 0, 1, 2, 3, 4
COST_VIA, COST_PHUBER = 0, 1
RO_NAN_TO_1E5, RO_ACCEPT_TEST, RO_ABSOLUTE = 1, 2, 4
PROJ_NONE, PROJ_BOX, PROJ_SETS = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "libisls_hip.so")


class View(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sb", C.c_int64), ("st", C.c_int64)]


class GainArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("solve_mode", C.c_int32), ("_pad", C.c_int32),
                ("A", View), ("Bm", View), ("Cxx", View), ("Cuu", View), ("Cux", View),
                ("K", C.c_void_p), ("Quu", C.c_void_p), ("fac", C.c_void_p), ("Qux", C.c_void_p),
                ("status", C.c_void_p), ("active", C.c_void_p), ("rec", C.c_void_p),
                ("lin_on", C.c_int32), ("lin_model", C.c_int32), ("lin_par", C.c_void_p), ("lin_par_sb", C.c_int64)]


class FfSeg(C.Structure):
    """Time-parallel form of the feed-forward pass (isls_ffseg): nseg <= 1 / G NULL = sequential."""
    _fields_ = [("nseg", C.c_int32), ("seg_len", C.c_int32), ("G", C.c_void_p), ("Psi", C.c_void_p), ("v", C.c_void_p)]


class FfArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("solve_mode", C.c_int32), ("_pad", C.c_int32),
                ("A", View), ("Bm", View), ("c0x", View), ("c0u", View), ("Qr", View), ("Rr", View),
                ("xhat", C.c_void_p), ("uhat", C.c_void_p),
                ("zx", C.c_void_p), ("lx", C.c_void_p), ("zu", C.c_void_p), ("lu", C.c_void_p),
                ("K", C.c_void_p), ("Quu", C.c_void_p), ("fac", C.c_void_p), ("Qux", C.c_void_p),
                ("k", C.c_void_p), ("active", C.c_void_p), ("seg", FfSeg), ("rec", C.c_void_p), ("Qr_term", C.c_void_p),
                ("lin_on", C.c_int32), ("lin_model", C.c_int32), ("lin_par", C.c_void_p), ("lin_par_sb", C.c_int64)]


class FfPrepareArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("solve_mode", C.c_int32), ("_pad", C.c_int32),
                ("A", View), ("Bm", View),
                ("K", C.c_void_p), ("Quu", C.c_void_p), ("fac", C.c_void_p), ("Qux", C.c_void_p),
                ("active", C.c_void_p), ("seg", FfSeg), ("rec", C.c_void_p)]


class RolloutArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("L", C.c_int32),
                ("model", C.c_int32), ("flags", C.c_int32), ("nvia", C.c_int32),
                ("model_par", C.c_void_p), ("model_par_sb", C.c_int64),
                ("K", C.c_void_p), ("k", C.c_void_p), ("xhat", C.c_void_p), ("uhat", C.c_void_p),
                ("x0", C.c_void_p), ("alphas", C.c_void_p),
                ("Qtab", C.c_void_p), ("Qtab_sb", C.c_int64), ("ztab", C.c_void_p), ("ztab_sb", C.c_int64),
                ("seq", C.c_void_p), ("q_nonzero", C.c_void_p), ("u_std", C.c_double),
                ("wq", View), ("wr", View),
                ("zx", C.c_void_p), ("lx", C.c_void_p), ("zu", C.c_void_p), ("lu", C.c_void_p),
                ("cost_cur", C.c_void_p), ("cost_all", C.c_void_p), ("best", C.c_void_p),
                ("cost_new", C.c_void_p), ("x_out", C.c_void_p), ("u_out", C.c_void_p),
                ("status", C.c_void_p), ("active", C.c_void_p),
                ("cost_model", C.c_int32), ("_pad2", C.c_int32), ("cost_par", C.c_void_p)]


class AdmmArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("proj_x", C.c_int32), ("proj_u", C.c_int32),
                ("relax", C.c_double), ("tol_abs", C.c_double), ("tol_rel", C.c_double),
                ("xx", C.c_void_p), ("xu", C.c_void_p),
                ("zx", C.c_void_p), ("lx", C.c_void_p), ("zu", C.c_void_p), ("lu", C.c_void_p),
                ("x_lo", View), ("x_hi", View), ("u_lo", View), ("u_hi", View),
                ("res", C.c_void_p), ("res_prev", C.c_void_p), ("active", C.c_void_p), ("iters", C.c_void_p),
                ("x_sets", C.c_void_p), ("u_sets", C.c_void_p), ("x_col0", C.c_int32), ("u_col0", C.c_int32),
                ("x_work", C.c_void_p), ("u_work", C.c_void_p)]


class ExpandArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("nvia", C.c_int32), ("_pad", C.c_int32),
                ("Qtab", C.c_void_p), ("Qtab_sb", C.c_int64), ("ztab", C.c_void_p), ("ztab_sb", C.c_int64),
                ("seq", C.c_void_p), ("u_std", C.c_double),
                ("Qr", View), ("Rr", View),
                ("xhat", C.c_void_p), ("uhat", C.c_void_p),
                ("Cxx", C.c_void_p), ("Cuu", C.c_void_p), ("c0x", C.c_void_p), ("c0u", C.c_void_p),
                ("cost", C.c_void_p), ("active", C.c_void_p),
                ("cost_model", C.c_int32), ("_pad2", C.c_int32), ("cost_par", C.c_void_p), ("q_nonzero", C.c_void_p)]


class LinearizeArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("model", C.c_int32), ("_pad", C.c_int32),
                ("model_par", C.c_void_p), ("model_par_sb", C.c_int64),
                ("xhat", C.c_void_p), ("uhat", C.c_void_p), ("A", C.c_void_p), ("Bm", C.c_void_p),
                ("active", C.c_void_p)]


class AcceptArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32),
                ("xx", C.c_void_p), ("xu", C.c_void_p), ("cost_new", C.c_void_p),
                ("xhat", C.c_void_p), ("uhat", C.c_void_p), ("cost", C.c_void_p),
                ("cost_hist", C.c_void_p), ("hist_len", C.c_void_p),
                ("tol_cost", C.c_double), ("tol_osc", C.c_double), ("outer_active", C.c_void_p)]


class CSet(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dim", C.c_int32), ("A", C.c_void_p), ("b", C.c_void_p), ("par", C.c_void_p),
                ("A_sp", C.c_int64), ("b_sp", C.c_int64), ("par_sp", C.c_int64)]


class ProjectArgs(C.Structure):
    _fields_ = [("P", C.c_int32), ("R", C.c_int32), ("d", C.c_int32), ("nsets", C.c_int32),
                ("max_iter", C.c_int32), ("algorithm", C.c_int32), ("rho", C.c_double), ("threshold", C.c_double),
                ("sets", CSet * 4),
                ("y_in", C.c_void_p), ("in_sp", C.c_int64), ("in_sr", C.c_int64),
                ("y_out", C.c_void_p), ("out_sp", C.c_int64), ("out_sr", C.c_int64),
                ("iters", C.c_void_p), ("active", C.c_void_p), ("row_mask", C.c_void_p), ("next", C.c_void_p)]


class SlsAdmmArgs(C.Structure):
    _fields_ = [("P", C.c_int32), ("R", C.c_int32), ("D", C.c_int32), ("max_iter", C.c_int32),
                ("alpha", C.c_double), ("tol", C.c_double), ("rel_tol", C.c_double),
                ("Linv", C.c_void_p), ("r_side", C.c_void_p), ("rr", C.c_void_p),
                ("proj", ProjectArgs),
                ("x_u", C.c_void_p), ("z", C.c_void_p), ("lmb", C.c_void_p), ("logs", C.c_void_p), ("iters", C.c_void_p)]


class ColumnsArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("C", C.c_int32), ("_pad", C.c_int32),
                ("A", View), ("Bm", View), ("Cuu", View), ("c0u", View), ("Rr", View),
                ("K", C.c_void_p), ("k", C.c_void_p), ("zu", C.c_void_p), ("lu", C.c_void_p),
                ("dx", C.c_void_p), ("du", C.c_void_p), ("active", C.c_void_p)]


class DenseLoopArgs(C.Structure):
    _fields_ = [("M", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("model", C.c_int32), ("_pad", C.c_int32),
                ("model_par", C.c_void_p), ("K", C.c_void_p), ("k", C.c_void_p), ("xhat", C.c_void_p), ("uhat", C.c_void_p),
                ("x0", C.c_void_p), ("x_log", C.c_void_p), ("u_log", C.c_void_p)]


class ColumnsAdmmArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("n", C.c_int32), ("m", C.c_int32), ("C", C.c_int32), ("phase", C.c_int32),
                ("relax", C.c_double), ("tol_abs", C.c_double), ("tol_rel", C.c_double),
                ("xx", C.c_void_p), ("xu", C.c_void_p),
                ("zx", C.c_void_p), ("lx", C.c_void_p), ("zu", C.c_void_p), ("lu", C.c_void_p),
                ("zx_prev", C.c_void_p), ("zu_prev", C.c_void_p), ("x_nom", C.c_void_p), ("u_nom", C.c_void_p),
                ("x_work", C.c_void_p), ("u_work", C.c_void_p), ("Qr", View), ("Rr", View),
                ("res", C.c_void_p), ("res_prev", C.c_void_p), ("active", C.c_void_p), ("iters", C.c_void_p)]


class OuterArgs(C.Structure):
    _fields_ = [("gain", GainArgs), ("ff", FfArgs), ("ro", RolloutArgs), ("admm", AdmmArgs),
                ("J", C.c_int32), ("skip_gain", C.c_int32), ("begin_done", C.c_int32), ("_pad", C.c_int32),
                ("log", C.c_void_p), ("outer_active", C.c_void_p), ("timing", C.c_void_p)]


class ColumnsIterationArgs(C.Structure):
    _fields_ = [("ff", FfArgs), ("cols", ColumnsArgs), ("ls", RolloutArgs), ("admm", ColumnsAdmmArgs),
                ("proj_x", C.c_void_p), ("proj_u", C.c_void_p), ("zero_x", C.c_void_p), ("zero_u", C.c_void_p), ("log", C.c_void_p),
                ("any_active", C.c_void_p)]


class AdvanceArgs(C.Structure):
    _fields_ = [("accept", AcceptArgs), ("lin", LinearizeArgs), ("exp", ExpandArgs),
                ("admm_active", C.c_void_p), ("iters", C.c_void_p), ("lx", C.c_void_p), ("lu", C.c_void_p), ("res_prev", C.c_void_p)]


# names every build of the library must export (checked by tests/test_capi_symbols.py)
EXPORTED = [f"isls_{k}_{s}" for s in ("f64", "f32") for k in
            ("riccati_gain", "riccati_ff", "riccati_gain_ff", "riccati_ff_prepare", "rollout_ls", "admm_update", "project_rows", "sls_admm", "sls_closed_loop", "columns_rollout", "columns_admm", "dense_closed_loop", "expand_quadratic", "linearize",
             "accept_step", "reduce_convergence", "reduce_convergence_table", "ilqr_admm_outer", "outer_advance", "columns_iteration")] + \
           ["isls_ff_segments", "isls_ff_record_elems", "isls_version", "isls_dims_supported", "isls_dims_generic", "isls_error_string", "isls_timing_create",
            "isls_timing_destroy", "isls_timing_reset", "isls_timing_pause", "isls_timing_read_ms"]


SET_BOX, SET_SOC_UNIT, SET_SQUARE, SET_LINEAR, SET_QUADRATIC, SET_SHELL, SET_MULTILINEAR = 1, 2, 3, 4, 5, 6, 7
ALG_ADMM, ALG_DYKSTRA, ALG_SOC = 0, 1, 2
MAX_ROW_DIM, MAX_SET_DIM, MAX_SETS = 4, 5, 4


class IslsError(RuntimeError):
    pass


def load_hip_library(path=None):
    """dlopen the HIP library; fail loudly (no CPU fallback exists by design)."""
    path = path or os.environ.get("ISLS_HIP_LIB") or HIP_LIB_PATH       # env override: A/B builds when tuning
    if not os.path.exists(path):
        raise IslsError(f"{path} not found: build it with `python __graft_entry__.py` "
                        f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.isls_version.restype = C.c_int
    if lib.isls_version() != ABI_VERSION:                      # a stale build would read the argument blocks with another layout
        raise IslsError(f"{path} reports ABI version {lib.isls_version()}, this binding is for {ABI_VERSION}: rebuild it "
                        f"(`python __graft_entry__.py`)")
    lib.isls_error_string.restype = C.c_char_p
    lib.isls_timing_create.restype = C.c_void_p
    lib.isls_timing_destroy.restype = None
    lib.isls_timing_destroy.argtypes = [C.c_void_p]
    lib.isls_timing_reset.argtypes = [C.c_void_p]
    lib.isls_timing_pause.argtypes = [C.c_void_p, C.c_int]
    lib.isls_timing_read_ms.restype = C.c_double
    lib.isls_timing_read_ms.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    return lib


_DIMS_LIB = None


def dims_supported(n, m):
    """True when the HIP library carries kernels for state dimension n and control dimension m (isls_dims_supported)."""
    global _DIMS_LIB
    if _DIMS_LIB is None:
        _DIMS_LIB = load_hip_library()
        _DIMS_LIB.isls_dims_supported.restype = C.c_int32
    return bool(_DIMS_LIB.isls_dims_supported(C.c_int32(int(n)), C.c_int32(int(m))))


def dims_generic(n, m):
    """True when (n, m) is served at all: by the templated kernels or by the generic ones of csrc/generic.hip (n <= 16, m <= 8)."""
    global _DIMS_LIB
    if _DIMS_LIB is None:
        _DIMS_LIB = load_hip_library()
    _DIMS_LIB.isls_dims_generic.restype = C.c_int32
    return bool(_DIMS_LIB.isls_dims_generic(C.c_int32(int(n)), C.c_int32(int(m))))


def supported_dims(n_max=16, m_max=8):
    """The (x_dim, u_dim) pairs the loaded library was built for."""
    return [(n, m) for n in range(1, n_max + 1) for m in range(1, m_max + 1) if dims_supported(n, m)]


# ------------------------------------------------------------------------------------------------
# array helpers (numpy or torch)
# ------------------------------------------------------------------------------------------------
def _is_torch(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return x.data_ptr()
    return x.ctypes.data


def _estrides(x):
    if _is_torch(x):
        return tuple(x.stride())
    return tuple(s // x.itemsize for s in x.strides)


def _sfx(x):
    dt = x.dtype
    if dt in (np.float64,) or (torch is not None and dt == torch.float64):
        return "f64"
    if dt in (np.float32,) or (torch is not None and dt == torch.float32):
        return "f32"
    raise TypeError(f"unsupported dtype {dt}")


def _dense(x, shape, name):
    """Check a dense C-contiguous operand of exactly `shape`."""
    if x is None:
        return None
    if tuple(x.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(x.shape)}")
    contiguous = x.is_contiguous() if _is_torch(x) else x.flags["C_CONTIGUOUS"]
    if not contiguous:
        raise ValueError(f"{name}: must be C-contiguous")
    return x


def ff_record_elems(B, N, n, m):
    """isls_ff_record_elems: elements of the packed-record buffer of the gain pass (blocked by wavefront)."""
    tpw = 64 // (n + m)
    model_words = 6 if (n, m) in ((9, 3), (4, 2)) else 0       # rec_model_words (csrc/isls_common.hpp): the arm's A[6:8, 0:3] / the car's six entries behind fac
    return -(-B // tpw) * tpw * N * ((n * n + 2 * n * m + m * m + model_words + 1) & ~1)   # record stride padded to an even word count


def _record(rec, B, N, n, m):
    if rec is None:
        return None
    size = rec.numel() if _is_torch(rec) else rec.size
    contiguous = rec.is_contiguous() if _is_torch(rec) else rec.flags["C_CONTIGUOUS"]
    if size < ff_record_elems(B, N, n, m) or not contiguous:
        raise ValueError(f"rec: needs a contiguous buffer of {ff_record_elems(B, N, n, m)} elements")
    return rec


def make_view(x, B, N, core, name):
    """Strided per-(b,t) operand: trailing dims == core (contiguous), leading dims broadcast to (B,N)."""
    if x is None:
        return View(None, 0, 0)
    core = tuple(core)
    nc = len(core)
    if tuple(x.shape[x.ndim - nc:]) != core:
        raise ValueError(f"{name}: trailing dims {tuple(x.shape)} != {core}")
    lead = tuple(x.shape[: x.ndim - nc])
    if len(lead) > 2:
        raise ValueError(f"{name}: at most two leading dims (batch, time)")
    st = _estrides(x)
    exp = 1
    for d, s in zip(reversed(core), reversed(st[x.ndim - nc:])):
        if d != 1 and s != exp:
            raise ValueError(f"{name}: trailing dims must be contiguous")
        exp *= d
    lead_st = st[: x.ndim - nc]
    if len(lead) == 0:
        sb = s_t = 0
    elif len(lead) == 1:            # [N, ...]: shared over the batch
        if lead[0] not in (1, N):
            raise ValueError(f"{name}: leading dim {lead[0]} is neither 1 nor N={N}")
        sb, s_t = 0, (0 if lead[0] == 1 else lead_st[0])
    else:
        if lead[0] not in (1, B) or lead[1] not in (1, N):
            raise ValueError(f"{name}: leading dims {lead} do not broadcast to ({B},{N})")
        sb = 0 if lead[0] == 1 else lead_st[0]
        s_t = 0 if lead[1] == 1 else lead_st[1]
    return View(_ptr(x), sb, s_t)


def _tab(x, B, core, name):
    """Per-batch table [core] or [B, core]: returns (ptr, batch stride in elements)."""
    core = tuple(core)
    if tuple(x.shape) == core:
        _dense(x, core, name)
        return _ptr(x), 0
    _dense(x, (B,) + core, name)
    return _ptr(x), int(np.prod(core))


class Kernels:
    """Thin marshaling layer over one shared library exporting `<prefix><kernel>_<f64|f32>`."""

    def __init__(self, lib, prefix="isls_", with_stream=True):
        self.lib, self.prefix, self.with_stream = lib, prefix, with_stream

    # -- plumbing ---------------------------------------------------------------------------------
    def _call(self, name, sfx, args, stream):
        fn = getattr(self.lib, f"{self.prefix}{name}_{sfx}")
        fn.restype = C.c_int
        if self.with_stream:
            rc = fn(C.byref(args), C.c_void_p(stream or 0))
        else:
            rc = fn(C.byref(args))
        if rc != OK:
            msg = {ERR_ARG: "bad argument", ERR_UNSUPPORTED: "unsupported (n,m)/model/L", ERR_LAUNCH: "launch failed"}
            raise IslsError(f"{self.prefix}{name}_{sfx} -> {rc} ({msg.get(rc, '?')})")
        return rc

    # -- argument builders (also used to fill OuterArgs) --------------------------------------------
    @staticmethod
    def _set_lin(a, lin, B, dtype):
        """hint fields of isls_gain_args / isls_ff_args: lin = (model id, parameters [P] or [B, P]) or None"""
        if lin is None:
            a.lin_on, a.lin_model, a.lin_par, a.lin_par_sb = 0, 0, None, 0
            return
        model, par = lin
        if par.dtype != dtype or not par.is_contiguous() or par.ndim not in (1, 2) or (par.ndim == 2 and par.shape[0] != B):
            raise ValueError("lin parameters: contiguous [P] or [B, P] of the pass's dtype")
        a.lin_on, a.lin_model, a.lin_par, a.lin_par_sb = 1, int(model), _ptr(par), (par.shape[1] if par.ndim == 2 else 0)

    @staticmethod
    def gain_args(A, Bm, Cxx, Cuu, K, Quu, fac, Qux, Cux=None, solve_mode=SOLVE_CHOL, status=None, active=None, rec=None, lin=None):
        """lin = (model id, parameters): A, Bm are isls_linearize's output for that model (isls_gain_args.lin_on)"""
        B, N, m, n = K.shape
        a = GainArgs(B=B, N=N, n=n, m=m, solve_mode=solve_mode)
        Kernels._set_lin(a, lin, B, K.dtype)
        a.A, a.Bm = make_view(A, B, N, (n, n), "A"), make_view(Bm, B, N, (n, m), "B")
        a.Cxx, a.Cuu = make_view(Cxx, B, N, (n, n), "Cxx"), make_view(Cuu, B, N, (m, m), "Cuu")
        a.Cux = make_view(Cux, B, N, (m, n), "Cux")
        a.K, a.Quu = _ptr(_dense(K, (B, N, m, n), "K")), _ptr(_dense(Quu, (B, N, m, m), "Quu"))
        a.fac, a.Qux = _ptr(_dense(fac, (B, N, m, m), "fac")), _ptr(_dense(Qux, (B, N, m, n), "Qux"))
        a.status, a.active = _ptr(status), _ptr(active)
        a.rec = _ptr(_record(rec, B, N, n, m))
        return a

    @staticmethod
    def ff_args(A, Bm, c0x, c0u, K, Quu, fac, Qux, k, Qr=None, Rr=None, xhat=None, uhat=None,
                zx=None, lx=None, zu=None, lu=None, solve_mode=SOLVE_CHOL, active=None, seg=None, rec=None, ncol=1, Qr_term=None,
                lin=None):
        """lin = (model id, parameters [P] or [B, P]): A, Bm are isls_linearize's output for that model (isls_ff_args.lin_on)"""
        B, N, m, n = K.shape
        a = FfArgs(B=B, N=N, n=n, m=m, solve_mode=solve_mode, _pad=int(ncol) if ncol and ncol > 1 else 0)
        if lin is not None and rec is None:
            raise ValueError("lin goes with the packed records")
        Kernels._set_lin(a, lin, B, K.dtype)
        if Qr_term is not None:                                 # weight block of the last step (isls_ff_args.Qr_term): [n,n]
            if Qr is None or tuple(Qr.shape) not in ((n, n), (1, n, n), (1, 1, n, n)):
                raise ValueError("Qr_term goes with a batch-shared, time-invariant Qr block")
            a.Qr_term = _ptr(_dense(Qr_term, (n, n), "Qr_term"))
        if ncol and ncol > 1:                                   # feedback columns: [C,B,N,.] blocks, checked as C*B trajectories
            col = lambda t, d: None if t is None else _dense(t, (ncol, B, N, d), "column block").reshape(ncol * B, N, d)   # noqa: E731
            zx, lx, zu, lu, k = col(zx, n), col(lx, n), col(zu, m), col(lu, m), col(k, m)
            a.zx, a.lx, a.zu, a.lu, a.k = _ptr(zx), _ptr(lx), _ptr(zu), _ptr(lu), _ptr(k)
            if Qr is not None and (zx is None or lx is None) or Rr is not None and (zu is None or lu is None):
                raise ValueError("Qr / Rr given without z / l")
            a.A, a.Bm = make_view(A, B, N, (n, n), "A"), make_view(Bm, B, N, (n, m), "B")
            a.c0x, a.c0u = make_view(c0x, B, N, (n,), "c0x"), make_view(c0u, B, N, (m,), "c0u")
            a.Qr, a.Rr = make_view(Qr, B, N, (n, n), "Qr"), make_view(Rr, B, N, (m, m), "Rr")
            a.xhat, a.uhat = _ptr(_dense(xhat, (B, N, n), "xhat")), _ptr(_dense(uhat, (B, N, m), "uhat"))
            a.K = _ptr(_dense(K, (B, N, m, n), "K"))
            a.active = _ptr(active)
            a.rec = _ptr(_record(rec, B, N, n, m))
            if seg is not None:
                a.seg = seg
            return a
        if seg is not None:
            a.seg = seg
        a.A, a.Bm = make_view(A, B, N, (n, n), "A"), make_view(Bm, B, N, (n, m), "B")
        a.c0x, a.c0u = make_view(c0x, B, N, (n,), "c0x"), make_view(c0u, B, N, (m,), "c0u")
        a.Qr, a.Rr = make_view(Qr, B, N, (n, n), "Qr"), make_view(Rr, B, N, (m, m), "Rr")
        if Qr is not None and (zx is None or lx is None):
            raise ValueError("Qr given without zx/lx")
        if Rr is not None and (zu is None or lu is None):
            raise ValueError("Rr given without zu/lu")
        a.xhat, a.uhat = _ptr(_dense(xhat, (B, N, n), "xhat")), _ptr(_dense(uhat, (B, N, m), "uhat"))
        a.zx, a.lx = _ptr(_dense(zx, (B, N, n), "zx")), _ptr(_dense(lx, (B, N, n), "lx"))
        a.zu, a.lu = _ptr(_dense(zu, (B, N, m), "zu")), _ptr(_dense(lu, (B, N, m), "lu"))
        a.K, a.Quu = _ptr(_dense(K, (B, N, m, n), "K")), _ptr(_dense(Quu, (B, N, m, m), "Quu"))
        a.fac, a.Qux = _ptr(_dense(fac, (B, N, m, m), "fac")), _ptr(_dense(Qux, (B, N, m, n), "Qux"))
        a.k = _ptr(_dense(k, (B, N, m), "k"))
        a.active = _ptr(active)
        a.rec = _ptr(_record(rec, B, N, n, m))
        return a

    @staticmethod
    def ff_seg(G, Psi, v, seg_len):
        """isls_ffseg over caller-owned buffers G[B,N,m,n], Psi[B,nseg,n,n], v[B,nseg,n]."""
        B, N, m, n = G.shape
        nseg = int(Psi.shape[1])
        _dense(G, (B, N, m, n), "seg.G"), _dense(Psi, (B, nseg, n, n), "seg.Psi"), _dense(v, (B, nseg, n), "seg.v")
        return FfSeg(nseg=nseg, seg_len=int(seg_len), G=_ptr(G), Psi=_ptr(Psi), v=_ptr(v))

    @staticmethod
    def ff_prepare_args(A, Bm, K, Quu, fac, Qux, seg, solve_mode=SOLVE_CHOL, active=None, rec=None):
        B, N, m, n = K.shape
        a = FfPrepareArgs(B=B, N=N, n=n, m=m, solve_mode=solve_mode, seg=seg)
        a.A, a.Bm = make_view(A, B, N, (n, n), "A"), make_view(Bm, B, N, (n, m), "B")
        a.K, a.Quu = _ptr(_dense(K, (B, N, m, n), "K")), _ptr(_dense(Quu, (B, N, m, m), "Quu"))
        a.fac, a.Qux = _ptr(_dense(fac, (B, N, m, m), "fac")), _ptr(_dense(Qux, (B, N, m, n), "Qux"))
        a.active = _ptr(active)
        a.rec = _ptr(_record(rec, B, N, n, m))
        return a

    @staticmethod
    def rollout_args(model, model_par, K, k, xhat, uhat, alphas, Qtab, ztab, seq, u_std, x_out, u_out,
                     best=None, cost_new=None, cost_all=None, x0=None, wq=None, wr=None, zx=None, lx=None,
                     zu=None, lu=None, cost_cur=None, flags=0, status=None, active=None, q_nonzero=None,
                     cost_model=COST_VIA, cost_par=None):
        B, N, m, n = K.shape
        L = int(alphas.shape[0])
        nvia = int(Qtab.shape[-3])
        a = RolloutArgs(B=B, N=N, n=n, m=m, L=L, model=model, flags=flags, nvia=nvia, u_std=float(u_std))
        if model_par.ndim == 1:
            a.model_par, a.model_par_sb = _ptr(model_par), 0
        else:
            _dense(model_par, (B, model_par.shape[1]), "model_par")
            a.model_par, a.model_par_sb = _ptr(model_par), model_par.shape[1]
        a.K, a.k = _ptr(_dense(K, (B, N, m, n), "K")), _ptr(_dense(k, (B, N, m), "k"))
        a.xhat, a.uhat = _ptr(_dense(xhat, (B, N, n), "xhat")), _ptr(_dense(uhat, (B, N, m), "uhat"))
        a.x0 = _ptr(_dense(x0, (B, n), "x0"))
        a.alphas = _ptr(alphas)
        a.Qtab, a.Qtab_sb = _tab(Qtab, B, (nvia, n, n), "Qtab")
        a.ztab, a.ztab_sb = _tab(ztab, B, (nvia, n), "ztab")
        if tuple(seq.shape) != (N,):
            raise ValueError("seq must be int32[N]")
        a.seq = _ptr(seq)
        if q_nonzero is not None and tuple(q_nonzero.shape) != (N,):
            raise ValueError("q_nonzero must be int32[N]")
        a.q_nonzero = _ptr(q_nonzero)
        a.wq, a.wr = make_view(wq, B, N, (n,), "wq"), make_view(wr, B, N, (m,), "wr")
        if wq is not None and (zx is None or lx is None):
            raise ValueError("wq given without zx/lx")
        if wr is not None and (zu is None or lu is None):
            raise ValueError("wr given without zu/lu")
        a.zx, a.lx = _ptr(_dense(zx, (B, N, n), "zx")), _ptr(_dense(lx, (B, N, n), "lx"))
        a.zu, a.lu = _ptr(_dense(zu, (B, N, m), "zu")), _ptr(_dense(lu, (B, N, m), "lu"))
        if (flags & RO_ACCEPT_TEST) and cost_cur is None:
            raise ValueError("ISLS_RO_ACCEPT_TEST needs cost_cur")
        a.cost_cur = _ptr(_dense(cost_cur, (B,), "cost_cur"))
        a.cost_all = _ptr(_dense(cost_all, (B, L), "cost_all"))
        a.best, a.cost_new = _ptr(best), _ptr(_dense(cost_new, (B,), "cost_new"))
        a.x_out, a.u_out = _ptr(_dense(x_out, (B, N, n), "x_out")), _ptr(_dense(u_out, (B, N, m), "u_out"))
        a.status, a.active = _ptr(status), _ptr(active)
        a.cost_model, a.cost_par = int(cost_model), _ptr(cost_par)
        return a

    @staticmethod
    def admm_args(xx, xu, res, zx=None, lx=None, zu=None, lu=None, x_lo=None, x_hi=None, u_lo=None, u_hi=None,
                  relax=1.0, tol_abs=0.0, tol_rel=0.0, res_prev=None, active=None, iters=None,
                  x_sets=None, x_col0=0, x_work=None, u_sets=None, u_col0=0, u_work=None):
        """x_sets / u_sets: a ProjectArgs descriptor (project_args) => ISLS_PROJ_SETS on that block, with the
        [B,N,.] scratch x_work / u_work and the first projected coordinate x_col0 / u_col0."""
        B, N, n = xx.shape
        m = xu.shape[2]
        a = AdmmArgs(B=B, N=N, n=n, m=m, relax=float(relax), tol_abs=float(tol_abs), tol_rel=float(tol_rel))
        a.proj_x = PROJ_SETS if x_sets is not None else (PROJ_BOX if x_lo is not None else PROJ_NONE)
        a.proj_u = PROJ_SETS if u_sets is not None else (PROJ_BOX if u_lo is not None else PROJ_NONE)
        if x_sets is not None:
            a.x_sets, a.x_col0, a.x_work = C.addressof(x_sets), int(x_col0), _ptr(_dense(x_work, (B, N, n), "x_work"))
        if u_sets is not None:
            a.u_sets, a.u_col0, a.u_work = C.addressof(u_sets), int(u_col0), _ptr(_dense(u_work, (B, N, m), "u_work"))
        a._keep = (x_sets, u_sets)
        a.xx, a.xu = _ptr(_dense(xx, (B, N, n), "xx")), _ptr(_dense(xu, (B, N, m), "xu"))
        a.zx, a.lx = _ptr(_dense(zx, (B, N, n), "zx")), _ptr(_dense(lx, (B, N, n), "lx"))
        a.zu, a.lu = _ptr(_dense(zu, (B, N, m), "zu")), _ptr(_dense(lu, (B, N, m), "lu"))
        a.x_lo, a.x_hi = make_view(x_lo, B, N, (n,), "x_lo"), make_view(x_hi, B, N, (n,), "x_hi")
        a.u_lo, a.u_hi = make_view(u_lo, B, N, (m,), "u_lo"), make_view(u_hi, B, N, (m,), "u_hi")
        a.res, a.res_prev = _ptr(_dense(res, (B, 2), "res")), _ptr(_dense(res_prev, (B, 2), "res_prev"))
        a.active, a.iters = _ptr(active), _ptr(iters)
        return a

    # -- kernels -------------------------------------------------------------------------------------
    def riccati_gain(self, *args, stream=None, **kw):
        a = self.gain_args(*args, **kw)
        return self._call("riccati_gain", _sfx(args[4]), a, stream)

    def riccati_ff(self, *args, stream=None, **kw):
        a = self.ff_args(*args, **kw)
        return self._call("riccati_ff", _sfx(args[4]), a, stream)

    def riccati_gain_ff(self, gain, ff, sfx, stream=None):
        """Gain pass + first feed-forward pass in one launch (argument blocks from gain_args / ff_args)."""
        fn = getattr(self.lib, f"isls_riccati_gain_ff_{sfx}")
        fn.restype = C.c_int
        rc = fn(C.byref(gain), C.byref(ff), C.c_void_p(stream or 0))
        if rc != OK:
            raise IslsError(f"isls_riccati_gain_ff_{sfx} -> {rc}: {self.lib.isls_error_string(rc).decode()}")

    @staticmethod
    def project_args(y_in, y_out, sets, rho=1.0, max_iter=200, threshold=1e-4, iters=None, active=None, cols=None,
                     algorithm=ALG_ADMM, row_mask=None, next_stage=None):
        """isls_project_args for rows y[P,R,D]; `cols=(c0, d)` projects the coordinate block [c0, c0+d) of every row
        (the rest of the row is not touched).  sets: list of dicts kind, dim, A[dim,d] | [P,dim,d], b, par (arrays of
        the dtype of y; a leading P axis gives per-problem operands)."""
        P, R, D = y_in.shape
        c0, d = cols if cols is not None else (0, D)
        if not (1 <= d <= MAX_ROW_DIM) or c0 < 0 or c0 + d > D or not (1 <= len(sets) <= MAX_SETS):
            raise ValueError("project_rows: unsupported row / set dimensions")
        _dense(y_in, (P, R, D), "y_in"), _dense(y_out, (P, R, D), "y_out")
        esz = y_in.element_size() if _is_torch(y_in) else y_in.itemsize
        a = ProjectArgs(P=P, R=R, d=d, nsets=len(sets), max_iter=int(max_iter), rho=float(rho), threshold=float(threshold),
                        algorithm=int(algorithm))
        if row_mask is not None:
            if tuple(row_mask.shape) != (R,):
                raise ValueError("row_mask must be int32[R]")
            a.row_mask = _ptr(row_mask)
        a.y_in, a.y_out = _ptr(y_in) + c0 * esz, _ptr(y_out) + c0 * esz
        a.in_sp = a.out_sp = R * D
        a.in_sr = a.out_sr = D
        keep = []
        for i, st in enumerate(sets):
            c = a.sets[i]
            c.kind, c.dim = int(st["kind"]), int(st["dim"])
            for name, core in (("A", (c.dim, d)), ("b", (c.dim,)), ("par", None)):
                arr = st.get(name)
                if arr is None:
                    continue
                per = arr.ndim == (len(core) + 1 if core is not None else 2)
                if core is not None:
                    _dense(arr, ((P,) + core) if per else core, f"sets[{i}].{name}")
                elif per and arr.shape[0] != P:
                    raise ValueError(f"sets[{i}].par: leading axis must be P")
                setattr(c, name, _ptr(arr))
                setattr(c, name + "_sp", int(arr[0].numel() if _is_torch(arr) else arr[0].size) if per else 0)
                keep.append(arr)
        a.iters, a.active = _ptr(iters), _ptr(active)
        if next_stage is not None:                               # a ProjectArgs applied in place to this stage's output
            a.next = C.addressof(next_stage)
        a._keep = keep + [row_mask, next_stage]
        a._spec = dict(sets=sets, rho=rho, max_iter=max_iter, threshold=threshold, cols=(c0, d), algorithm=int(algorithm),
                       row_mask=row_mask, next_stage=next_stage)                                 # to rebuild elsewhere
        return a

    @staticmethod
    def project_args_chain(y_in, y_out, stages, wrap=lambda a: a, iters=None, active=None):
        """Descriptor of a `projections.ConvexSets` with all its stages (`.then` chain): the stages run in order, in place on
        the first stage's output.  `wrap` moves a numpy operand to where the library expects it (device tensor / identity)."""
        R = y_in.shape[1]
        nxt = None
        for k in range(len(stages) - 1, -1, -1):
            cs = stages[k]
            if cs.cols != stages[0].cols:
                raise ValueError("the stages of a ConvexSets chain must act on the same coordinate block")
            sets = [{k_: (wrap(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k_, v in st.items()} for st in cs.sets]
            mask = cs.row_mask(R)
            nxt = Kernels.project_args(y_in if k == 0 else y_out, y_out, sets, rho=cs.rho, max_iter=cs.max_iter, threshold=cs.threshold,
                                       iters=iters if k == 0 else None, active=active, cols=cs.cols, algorithm=cs.algorithm,
                                       row_mask=None if mask is None else wrap(mask), next_stage=nxt)
        return nxt

    def project_rows(self, y_in, y_out, sets, stream=None, **kw):
        a = self.project_args(y_in, y_out, sets, **kw)
        return self._call("project_rows", _sfx(y_in), a, stream)

    def sls_admm(self, Linv, r_side, rr, sets, x_u, alpha=1.0, tol=1e-3, max_iter=50, rho=1.0, inner_max_iter=200,
                 threshold=1e-4, z=None, lmb=None, logs=None, iters=None, rel_tol=1e-2, stream=None):
        """isls_sls_admm: Linv [R,R], r_side [P,R,D], rr [R], sets as in project_args (A [dim,D] or [P,dim,D] ...)."""
        P, R, D = r_side.shape
        _dense(Linv, (R, R), "Linv"), _dense(rr, (R,), "rr"), _dense(x_u, (P, R, D), "x_u")
        pa = self.project_args(x_u, x_u, sets, rho=rho, max_iter=inner_max_iter, threshold=threshold)
        a = SlsAdmmArgs(P=P, R=R, D=D, max_iter=int(max_iter), alpha=float(alpha), tol=float(tol), rel_tol=float(rel_tol),
                        proj=pa)
        a.Linv, a.r_side, a.rr = _ptr(Linv), _ptr(_dense(r_side, (P, R, D), "r_side")), _ptr(rr)
        a.x_u, a.z, a.lmb = _ptr(x_u), _ptr(_dense(z, (P, R, D), "z")), _ptr(_dense(lmb, (P, R, D), "lmb"))
        a.logs, a.iters = _ptr(_dense(logs, (P, int(max_iter), 2), "logs")), _ptr(iters)
        a._keep = pa
        return self._call("sls_admm", _sfx(r_side), a, stream)

    def dense_closed_loop(self, model, model_par, K, k, x0, x_log, u_log, xhat=None, uhat=None, stream=None):
        """isls_dense_closed_loop: K [N m, N n], k [N m], x0 [M,n], nominal xhat [N,n] / uhat [N,m] of one problem."""
        M, N, n = x_log.shape
        m = u_log.shape[2]
        a = DenseLoopArgs(M=M, N=N, n=n, m=m, model=int(model))
        a.model_par, a.K, a.k = _ptr(model_par), _ptr(_dense(K, (N * m, N * n), "K")), _ptr(_dense(k, (N * m,), "k"))
        a.xhat, a.uhat = _ptr(_dense(xhat, (N, n), "xhat")), _ptr(_dense(uhat, (N, m), "uhat"))
        a.x0, a.x_log, a.u_log = _ptr(_dense(x0, (M, n), "x0")), _ptr(x_log), _ptr(_dense(u_log, (M, N, m), "u_log"))
        return self._call("dense_closed_loop", _sfx(x_log), a, stream)

    def columns_rollout(self, *args, stream=None, **kw):
        return self._call("columns_rollout", _sfx(args[6]), self.columns_args(*args, **kw), stream)

    @staticmethod
    def columns_args(A, Bm, Cuu, c0u, K, k, dx, du, Rr=None, zu=None, lu=None, active=None):
        """isls_columns_rollout: k, dx, du column-major [C,B,N,.]; A, Bm, Cuu, c0u, Rr broadcastable views."""
        Cc, B, N, n = dx.shape
        m = du.shape[3]
        a = ColumnsArgs(B=B, N=N, n=n, m=m, C=Cc)
        a.A, a.Bm = make_view(A, B, N, (n, n), "A"), make_view(Bm, B, N, (n, m), "B")
        a.Cuu, a.c0u = make_view(Cuu, B, N, (m, m), "Cuu"), make_view(c0u, B, N, (m,), "c0u")
        a.Rr = make_view(Rr, B, N, (m, m), "Rr")
        a.K, a.k = _ptr(_dense(K, (B, N, m, n), "K")), _ptr(_dense(k, (Cc, B, N, m), "k"))
        a.zu, a.lu = _ptr(_dense(zu, (Cc, B, N, m), "zu")), _ptr(_dense(lu, (Cc, B, N, m), "lu"))
        a.dx, a.du = _ptr(_dense(dx, (Cc, B, N, n), "dx")), _ptr(_dense(du, (Cc, B, N, m), "du"))
        a.active = _ptr(active)
        return a

    def columns_admm(self, phase, dims, res, res_prev, x=None, u=None, stream=None, **kw):
        a = self.columns_admm_args(phase, dims, res, res_prev, x=x, u=u, **kw)
        return self._call("columns_admm", _sfx((x or u)["xx"]), a, stream)

    @staticmethod
    def columns_admm_args(phase, dims, res, res_prev, x=None, u=None, relax=1.0, tol_abs=0.0, tol_rel=1e-3, active=None, iters=None):
        """isls_columns_admm; dims = (B, N, n, m, C); x / u: dict(xx, z, l, z_prev, work, W, nom=None) or None."""
        B, N, n, m, Cc = dims
        a = ColumnsAdmmArgs(B=B, N=N, n=n, m=m, C=Cc, phase=int(phase), relax=float(relax), tol_abs=float(tol_abs),
                            tol_rel=float(tol_rel))
        for blk, d, names in ((x, n, ("xx", "zx", "lx", "zx_prev", "x_nom", "x_work", "Qr")),
                              (u, m, ("xu", "zu", "lu", "zu_prev", "u_nom", "u_work", "Rr"))):
            if blk is None:
                continue
            for key, name in zip(("xx", "z", "l", "z_prev"), names[:4]):
                setattr(a, name, _ptr(_dense(blk[key], (Cc, B, N, d), name)))
            setattr(a, names[4], _ptr(_dense(blk.get("nom"), (B, N, d), names[4])))
            setattr(a, names[5], _ptr(_dense(blk["work"], (B, N * d, Cc), names[5])))
            setattr(a, names[6], make_view(blk["W"], B, N, (d, d), names[6]))
        a.res, a.res_prev = _ptr(_dense(res, (B, 2), "res")), _ptr(_dense(res_prev, (B, 2), "res_prev"))
        a.active, a.iters = _ptr(active), _ptr(iters)
        return a

    @staticmethod
    def columns_iteration_args(ff, cols, ls, admm, proj_x=None, proj_u=None, zero_x=None, zero_u=None, log=None):
        """isls_columns_iteration_args from the blocks of ff_args / columns_args / rollout_args (None: no line search) /
        columns_admm_args (None: unconstrained) and the row-projection descriptors of the two blocks."""
        a = ColumnsIterationArgs(ff=ff, cols=cols)
        if ls is not None:
            a.ls = ls
        if admm is not None:
            a.admm = admm
        a.proj_x = C.addressof(proj_x) if proj_x is not None else None
        a.proj_u = C.addressof(proj_u) if proj_u is not None else None
        a.zero_x, a.zero_u, a.log = _ptr(zero_x), _ptr(zero_u), _ptr(log)
        a._keep = (ff, cols, ls, admm, proj_x, proj_u)
        return a

    def columns_iteration(self, it, sfx, stream=None):
        return self._call("columns_iteration", sfx, it, stream)

    def sls_closed_loop(self, A, Bm, K, k, x0, x_log, u_log, stream=None):
        M, N, n = x_log.shape
        m = u_log.shape[2]
        _dense(A, (n, n), "A"), _dense(Bm, (n, m), "B"), _dense(K, (N * m, N * n), "K"), _dense(k, (N * m,), "k")
        _dense(x0, (M, n), "x0"), _dense(u_log, (M, N, m), "u_log")
        fn = getattr(self.lib, f"{self.prefix}sls_closed_loop_{_sfx(x_log)}")
        fn.restype = C.c_int
        args = [C.c_int32(M), C.c_int32(N), C.c_int32(n), C.c_int32(m)] + [C.c_void_p(_ptr(t)) for t in (A, Bm, K, k, x0, x_log, u_log)]
        rc = fn(*args, C.c_void_p(stream or 0)) if self.with_stream else fn(*args)
        if rc != OK:
            raise IslsError(f"{self.prefix}sls_closed_loop -> {rc}")
        return rc

    def riccati_ff_prepare(self, *args, stream=None, **kw):
        a = self.ff_prepare_args(*args, **kw)
        return self._call("riccati_ff_prepare", _sfx(args[2]), a, stream)

    def ff_segments(self, N, nseg_requested):
        """(nseg, seg_len) the library uses for a horizon of N steps."""
        fn = self.lib.isls_ff_segments
        fn.restype = C.c_int32
        seg_len = C.c_int32(0)
        nseg = int(fn(C.c_int32(int(N)), C.c_int32(int(nseg_requested)), C.byref(seg_len)))
        return nseg, int(seg_len.value)

    def rollout_ls(self, *args, stream=None, **kw):
        a = self.rollout_args(*args, **kw)
        return self._call("rollout_ls", _sfx(args[2]), a, stream)

    def admm_update(self, *args, stream=None, **kw):
        a = self.admm_args(*args, **kw)
        return self._call("admm_update", _sfx(args[0]), a, stream)

    def expand_quadratic(self, *args, stream=None, **kw):
        return self._call("expand_quadratic", _sfx(args[4]), self.expand_args(*args, **kw), stream)

    @staticmethod
    def expand_args(Qtab, ztab, seq, u_std, c0x, c0u, xhat=None, uhat=None, Cxx=None, Cuu=None,
                    Qr=None, Rr=None, cost=None, active=None, cost_model=COST_VIA, cost_par=None, q_nonzero=None):
        B, N, n = c0x.shape
        m = c0u.shape[2]
        nvia = int(Qtab.shape[-3])
        a = ExpandArgs(B=B, N=N, n=n, m=m, nvia=nvia, u_std=float(u_std))
        a.Qtab, a.Qtab_sb = _tab(Qtab, B, (nvia, n, n), "Qtab")
        a.ztab, a.ztab_sb = _tab(ztab, B, (nvia, n), "ztab")
        a.seq = _ptr(seq)
        a.Qr, a.Rr = make_view(Qr, B, N, (n, n), "Qr"), make_view(Rr, B, N, (m, m), "Rr")
        a.xhat, a.uhat = _ptr(_dense(xhat, (B, N, n), "xhat")), _ptr(_dense(uhat, (B, N, m), "uhat"))
        a.Cxx, a.Cuu = _ptr(_dense(Cxx, (B, N, n, n), "Cxx")), _ptr(_dense(Cuu, (B, N, m, m), "Cuu"))
        a.c0x, a.c0u = _ptr(_dense(c0x, (B, N, n), "c0x")), _ptr(_dense(c0u, (B, N, m), "c0u"))
        a.cost, a.active = _ptr(_dense(cost, (B,), "cost")), _ptr(active)
        a.cost_model, a.cost_par, a.q_nonzero = int(cost_model), _ptr(cost_par), _ptr(q_nonzero)
        return a

    def linearize(self, *args, stream=None, **kw):
        return self._call("linearize", _sfx(args[2]), self.linearize_args(*args, **kw), stream)

    @staticmethod
    def linearize_args(model, model_par, xhat, uhat, A, Bm, active=None):
        B, N, n = xhat.shape
        m = uhat.shape[2]
        a = LinearizeArgs(B=B, N=N, n=n, m=m, model=model)
        if model_par.ndim == 1:
            a.model_par, a.model_par_sb = _ptr(model_par), 0
        else:
            a.model_par, a.model_par_sb = _ptr(model_par), model_par.shape[1]
        a.xhat, a.uhat = _ptr(_dense(xhat, (B, N, n), "xhat")), _ptr(_dense(uhat, (B, N, m), "uhat"))
        a.A, a.Bm = _ptr(_dense(A, (B, N, n, n), "A")), _ptr(_dense(Bm, (B, N, n, m), "B"))
        a.active = _ptr(active)
        return a

    def accept_step(self, *args, stream=None, **kw):
        return self._call("accept_step", _sfx(args[0]), self.accept_args(*args, **kw), stream)

    @staticmethod
    def accept_args(xx, xu, cost_new, xhat, uhat, cost, cost_hist=None, hist_len=None, tol_cost=-1.0,
                    tol_osc=-1.0, outer_active=None):
        B, N, n = xx.shape
        m = xu.shape[2]
        a = AcceptArgs(B=B, N=N, n=n, m=m, tol_cost=float(tol_cost), tol_osc=float(tol_osc))
        a.xx, a.xu = _ptr(_dense(xx, (B, N, n), "xx")), _ptr(_dense(xu, (B, N, m), "xu"))
        a.cost_new = _ptr(_dense(cost_new, (B,), "cost_new"))
        a.xhat, a.uhat = _ptr(_dense(xhat, (B, N, n), "xhat")), _ptr(_dense(uhat, (B, N, m), "uhat"))
        a.cost = _ptr(_dense(cost, (B,), "cost"))
        a.cost_hist, a.hist_len = _ptr(_dense(cost_hist, (B, 8), "cost_hist")), _ptr(hist_len)
        a.outer_active = _ptr(outer_active)
        return a

    @staticmethod
    def advance_args(accept, lin=None, exp=None, admm_active=None, iters=None, lx=None, lu=None, res_prev=None):
        """isls_advance_args from the blocks of accept_args / linearize_args / expand_args (lin / exp None: that stage is skipped)"""
        a = AdvanceArgs(accept=accept)
        if lin is not None:
            a.lin = lin
        if exp is not None:
            a.exp = exp
        a.admm_active, a.iters, a.lx, a.lu, a.res_prev = _ptr(admm_active), _ptr(iters), _ptr(lx), _ptr(lu), _ptr(res_prev)
        a._keep = (accept, lin, exp)
        return a

    def outer_advance(self, adv, sfx, stream=None):
        return self._call("outer_advance", sfx, adv, stream)

    def reduce_convergence(self, cost, res, active, status, out5, stream=None):
        fn = getattr(self.lib, f"{self.prefix}reduce_convergence_{_sfx(out5)}")
        fn.restype = C.c_int
        B = int(cost.shape[0]) if cost is not None else int(res.shape[0])
        argv = [C.c_int32(B), C.c_void_p(_ptr(cost)), C.c_void_p(_ptr(res)), C.c_void_p(_ptr(active)),
                C.c_void_p(_ptr(status)), C.c_void_p(_ptr(out5))]
        if self.with_stream:
            argv.append(C.c_void_p(stream or 0))
        rc = fn(*argv)
        if rc != OK:
            raise IslsError(f"reduce_convergence -> {rc}")
        return rc

    def reduce_convergence_table(self, cost, res, active, status, table, rank, stream=None):
        """isls_reduce_convergence_table: the shard's five numbers into row `rank` of the [W,5] table, other rows zeroed."""
        world = int(table.shape[0])
        _dense(table, (world, 5), "table")
        fn = getattr(self.lib, f"{self.prefix}reduce_convergence_table_{_sfx(table)}")
        fn.restype = C.c_int
        B = int(cost.shape[0]) if cost is not None else int(res.shape[0])
        argv = [C.c_int32(B), C.c_void_p(_ptr(cost)), C.c_void_p(_ptr(res)), C.c_void_p(_ptr(active)),
                C.c_void_p(_ptr(status)), C.c_void_p(_ptr(table)), C.c_int32(int(rank)), C.c_int32(world)]
        if self.with_stream:
            argv.append(C.c_void_p(stream or 0))
        rc = fn(*argv)
        if rc != OK:
            raise IslsError(f"reduce_convergence_table -> {rc}")
        return rc

    def outer(self, gain, ff, ro, admm, J, sfx, skip_gain=False, log=None, outer_active=None, begin_done=False, stream=None):
        a = OuterArgs(gain=gain, ff=ff, ro=ro, admm=admm, J=int(J), skip_gain=int(bool(skip_gain)), begin_done=int(bool(begin_done)))
        a.log, a.outer_active = _ptr(log), _ptr(outer_active)
        return self._call("ilqr_admm_outer", sfx, a, stream)
