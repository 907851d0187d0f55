"""Shared state of the batched `iSLS` / `SLS` front ends (reference: isls/base.py, isls/isls_base.py,
isls/sls_base.py).  The numerics live in `engine.Engine` (HIP kernels); this layer only converts between the
reference's numpy-on-host calling conventions and the engine's device tensors.

Batch convention: `batch=1` behaves like the reference (arrays without a batch axis in and out); `batch=B`
adds a leading axis of size B to every per-trajectory quantity (x_nom [B,N,n], cost [B], K [B,N,m,n] ...).
Shared quantities (A,B of an LTI model, Qs, seq, bounds, rho) may be given without the batch axis.
"""
import numpy as np
import torch

from . import _capi as capi
from .engine import ALPHAS, Engine
from .projections import Box, ConvexSets, identify_box
from .utils import find_mus, find_precs


class Base:
    def __init__(self, x_dim, u_dim, N, batch=1, dtype=np.float64, device="cuda"):
        self.N, self.x_dim, self.u_dim, self.batch = int(N), int(x_dim), int(u_dim), int(batch)
        self.np_dtype = np.dtype(dtype)
        tdtype = torch.float64 if self.np_dtype == np.float64 else torch.float32
        self.engine = Engine(self.batch, self.N, self.x_dim, self.u_dim, dtype=tdtype, device=device)
        self.A = self.B = None
        self.zs = self.Qs = self.seq = self.Rt = None
        self.Q = self.xd = self.R = None

    # ---- conversions -------------------------------------------------------------------------------------
    def _out(self, t):
        """device tensor -> numpy, dropping the batch axis when batch == 1 (reference shapes)."""
        a = t.detach().cpu().numpy()
        return a[0] if self.batch == 1 else a

    def _batched(self, x, core_ndim):
        """numpy input -> [B, ...core] float array (adds / broadcasts the batch axis)."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == core_ndim:
            x = np.broadcast_to(x[None], (self.batch,) + x.shape)
        return np.ascontiguousarray(x)

    # ---- reference API -------------------------------------------------------------------------------------
    def compute_Rr_Qr(self, rho_x, rho_u, dp=True):
        """rho -> per-step weights (isls/base.py:55-79, dp=True form): Qr [N,n,n] array, Rr list of N [m,m]."""
        if not dp:
            raise NotImplementedError("dense block-diagonal weights belong to the batch-form solvers (out of scope)")

        def expand(rho, d):
            if rho is None:
                return None
            if isinstance(rho, (int, float)):
                return np.tile(float(rho) * np.eye(d)[None], (self.N, 1, 1))
            rho = np.asarray(rho, dtype=np.float64)
            return np.tile(rho[None], (self.N, 1, 1)) if rho.ndim == 2 else rho
        Qr, Rr = expand(rho_x, self.x_dim), expand(rho_u, self.u_dim)
        return Qr, (None if Rr is None else list(Rr))

    def set_quadratic_cost(self, zs, Qs, seq, u_std):
        """Via-point cost (isls/base.py:81-89): cost = sum_t (x_t-z_t)'Q_t(x_t-z_t) + u_std |u_t|^2, Q_t = Qs[seq[t]]."""
        self.zs, self.Qs = np.asarray(zs, dtype=np.float64), np.asarray(Qs, dtype=np.float64)
        self.seq = np.asarray(seq).astype(np.int32)
        self.u_std = float(u_std)
        self.Rt = np.eye(self.u_dim) * u_std
        self.Q = find_precs(self.Qs, self.seq)                 # [N,n,n] blocks of the reference's sparse Q
        if self.zs.ndim == 2:
            self.xd = find_mus(self.zs, self.seq)
        self.R = self.Rt
        self.engine.set_quadratic_cost(self.zs, self.Qs, self.seq, u_std)

    set_cost_variables = set_quadratic_cost                     # notebook-era name

    def _projection(self, project, d):
        """project_x / project_u argument -> Box descriptor (device path), callable (host path) or None."""
        if project is False or project is None:
            return None
        if isinstance(project, (Box, ConvexSets)):
            if isinstance(project, ConvexSets) and project.dim != d:
                raise ValueError(f"ConvexSets acts on rows of dimension {project.dim}, this block has {d}")
            return project
        if callable(project):
            box = identify_box(project, self.N * d)
            return box if box is not None else project
        raise TypeError("project_x / project_u must be False, a projections.Box or a callable")
