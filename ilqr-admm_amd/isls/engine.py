"""Device-side state and kernel sequencing of the batched DP-form iLQR-ADMM solver.

`Engine` owns the HBM-resident arrays of B independent trajectories (torch tensors on one MI355X) and
drives the HIP kernels of csrc/ through the C ABI (include/isls_hip.h).  It is the batched counterpart
of the state the reference keeps on an `iSLS`/`SLS` object (x_nom, u_nom, A, B, K, k, cost, z, lambda:
isls/isls_base.py:4-27, isls/base.py:11-24) and of the bodies of `iSLS.ilqr_admm` (isls/isls.py:420-499)
and `ADMM` (isls/admm.py:6-106).  torch is used for memory and streams only.

HBM layout: every array is [B, N, ...] row-major (trajectory-major), so one trajectory's horizon is a
contiguous stream (A: N*n*n, K: N*m*n, x: N*n ... elements) that a wavefront slot walks backwards
(Riccati) or forwards (rollout) with one-step-ahead prefetch; shared tables (LTI A/B, via-point Q, box
bounds, rho weights) are passed with zero batch/time strides and stay cache-resident.
"""
import ctypes
import os

import numpy as np
import torch

from . import _capi as capi

_LIB = None
_KERN = None


def kernels():
    """The process-wide binding of csrc/libisls_hip.so (raises if it has not been built)."""
    global _LIB, _KERN
    if _KERN is None:
        _LIB = capi.load_hip_library()
        _KERN = capi.Kernels(_LIB, prefix="isls_", with_stream=True)
    return _KERN


def library():
    kernels()
    return _LIB


ALPHAS = 10.0 ** np.linspace(0.0, -5.0, 50)            # line-search grid, isls/isls_base.py:10-11


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


class _Timed:
    def __init__(self, eng, name):
        self.eng, self.name = eng, name

    def __enter__(self):
        ev = getattr(self.eng, "profile_events", None)
        if ev is not None:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        ev = getattr(self.eng, "profile_events", None)
        if ev is not None:
            self.b.record()
            ev.setdefault(self.name, []).append((self.a, self.b))
        return False


class Engine:
    profile_events = None

    def __init__(self, batch, N, x_dim, u_dim, dtype=torch.float64, device="cuda"):
        if not torch.cuda.is_available():
            raise capi.IslsError("isls.Engine needs a HIP device (torch.cuda.is_available() is False); "
                                 "there is no CPU fallback")
        self.kern = kernels()
        self.B, self.N, self.n, self.m = int(batch), int(N), int(x_dim), int(u_dim)
        # pairs with an instantiation of the row-per-lane kernels run those; any other pair with n <= 16, m <= 8 runs the generic
        # kernels (csrc/generic.hip: array form, no packed records, no time-parallel segments); beyond that the library refuses
        self.fast_dims = capi.dims_supported(self.n, self.m)
        if not self.fast_dims and not capi.dims_generic(self.n, self.m):
            raise capi.IslsError(f"libisls_hip.so has no kernels for x_dim={self.n}, u_dim={self.m}: the generic kernels serve "
                                 f"x_dim <= 16, u_dim <= 8; instantiated fast pairs: {capi.supported_dims()}")
        self.dtype, self.device = dtype, torch.device(device)
        self.sfx = "f64" if dtype == torch.float64 else "f32"
        B, N, n, m = self.B, self.N, self.n, self.m
        z = lambda *s: torch.zeros(*s, dtype=dtype, device=self.device)          # noqa: E731
        zi = lambda *s: torch.zeros(*s, dtype=torch.int32, device=self.device)   # noqa: E731
        # nominal trajectory and its cost
        self.xhat, self.uhat, self.cost = z(B, N, n), z(B, N, m), z(B)
        # linearisation and cost expansion
        self.A, self.Bm = z(B, N, n, n), z(B, N, n, m)
        self.Cxx, self.Cuu, self.c0x, self.c0u = z(B, N, n, n), z(B, N, m, m), z(B, N, n), z(B, N, m)
        self.Cux = None
        # Riccati factors and gains
        self.K, self.Quu, self.fac, self.Qux, self.k = z(B, N, m, n), z(B, N, m, m), z(B, N, m, m), z(B, N, m, n), z(B, N, m)
        # x-step result (line-search winner)
        self.xx, self.xu, self.cost_new = z(B, N, n), z(B, N, m), z(B)
        self.best, self.status = zi(B), zi(B)
        # ADMM state
        self.zx = self.lx = self.zu = self.lu = None
        self.res, self.res_prev = z(B, 2), torch.full((B, 2), 1e6, dtype=dtype, device=self.device)
        self.outer_active = torch.ones(B, dtype=torch.int32, device=self.device)
        self.admm_active = torch.ones(B, dtype=torch.int32, device=self.device)
        self.admm_iters = zi(B)                              # executed ADMM iterations of the current outer iteration
        self.out5 = z(5)
        self.cost_hist, self.hist_len = z(B, 8), zi(B)      # tail of cost_log per trajectory (isls_base.py:85)
        self.alphas = torch.as_tensor(ALPHAS, dtype=dtype, device=self.device)
        # problem description
        self.model = None
        self.model_par = None
        self.Qtab = self.ztab = self.seq = self.q_nonzero = None
        self.u_std = 0.0
        self.cost_model, self.cost_par = capi.COST_VIA, None
        self.allow_shared_hessian = True                      # front ends that write Cxx / Cuu themselves switch it off
        self.Qr = self.Rr = self.wq = self.wr = None
        self.x_lo = self.x_hi = self.u_lo = self.u_hi = None
        self.x_sets = self.u_sets = self.x_work = self.u_work = None
        self.x_col0 = self.u_col0 = 0
        self.relax = 1.0
        self.solve_mode = capi.SOLVE_CHOL
        self._outer_args = None

    # ---- optional per-kernel-family event timing (bench.py) -------------------------------------------------
    def timed(self, name):
        """Context manager: with `self.profile_events` set to a dict, HIP events are recorded on the launch stream around
        the enclosed launches and collected under `name` (the kernels run on torch's current stream); otherwise a no-op."""
        return _Timed(self, name)

    def family_ms(self):
        """{name: (total ms, launches groups)} of the events recorded since profile_events was set; synchronises."""
        torch.cuda.synchronize()
        return {k: (sum(a.elapsed_time(b) for a, b in v), len(v)) for k, v in (self.profile_events or {}).items()}

    # ---- problem setup ---------------------------------------------------------------------------------
    def _t(self, x):
        if not isinstance(x, torch.Tensor):
            x = np.asarray(x)
            if not x.flags.writeable:                          # broadcast views: torch wants an array it may alias
                x = x.copy()
        return torch.as_tensor(x, dtype=self.dtype, device=self.device).contiguous()

    def set_model(self, model_id, par):
        """Built-in forward model (ISLS_MODEL_*); par is [P] (shared) or [B,P] (per trajectory)."""
        self.model, self.model_par = int(model_id), self._t(par)
        self._outer_args = None
        self._ab_made = self._ab_static = None                 # A, Bm no longer belong to the model in use

    def set_quadratic_cost(self, zs, Qs, seq, u_std):
        """Via-point quadratic cost (Base.set_quadratic_cost, isls/base.py:81-89); zs [nvia,n] or [B,nvia,n]."""
        self.ztab, self.Qtab = self._t(zs), self._t(Qs)
        seq = np.asarray(seq.cpu() if isinstance(seq, torch.Tensor) else seq).astype(np.int32)
        self.seq = torch.as_tensor(seq, device=self.device)
        qnz = (self.Qtab.reshape(-1, self.Qtab.shape[-3], self.n * self.n) != 0).any(-1).any(0).cpu().numpy()
        self.q_nonzero = torch.as_tensor(qnz[seq].astype(np.int32), device=self.device)
        self.u_std = float(u_std)
        self.cost_model, self.cost_par = capi.COST_VIA, None
        self._outer_args = None
        self._hess_dirty = True

    def set_cost_model(self, cost_model, par):
        """Built-in non-quadratic cost of the line search and of the expansion (ISLS_COST_PHUBER: [cu, cx, px, cf, pf])."""
        self.cost_model, self.cost_par = int(cost_model), self._t(np.ascontiguousarray(par))
        # the via-point tables are ignored by this cost model; one all-zero entry keeps every argument block well formed
        self.Qtab, self.ztab = torch.zeros(1, self.n, self.n, dtype=self.dtype, device=self.device), torch.zeros(1, self.n, dtype=self.dtype, device=self.device)
        self.seq = torch.zeros(self.N, dtype=torch.int32, device=self.device)
        self.q_nonzero = torch.zeros(self.N, dtype=torch.int32, device=self.device)
        self.u_std = 0.0
        self._outer_args = None

    def set_nominal(self, x_nom, u_nom):
        """nominal_values setter (isls/isls_base.py:80-85): stores the nominal and evaluates its cost."""
        self.xhat.copy_(self._t(x_nom).expand(self.B, self.N, self.n))
        self.uhat.copy_(self._t(u_nom).expand(self.B, self.N, self.m))
        self.evaluate_cost()
        self.cost_hist.zero_()
        self.cost_hist[:, 0] = self.cost                  # cost_log = [initial cost]
        self.hist_len.fill_(1)
        self.outer_active.fill_(1)
        self.status.zero_()

    def set_admm(self, rho_x=None, rho_u=None, x_box=None, u_box=None, relax=1.0, x_sets=None, u_sets=None):
        """ADMM weights (Base.compute_Rr_Qr, isls/base.py:55-79, dp=True form) and the constraint sets: boxes
        (lo, hi) or `isls.projections.ConvexSets` (project_set_convex over the time steps, on the device)."""
        B, N, n, m = self.B, self.N, self.n, self.m
        z = lambda *s: torch.zeros(*s, dtype=self.dtype, device=self.device)      # noqa: E731

        def weights(rho, d):
            if rho is None:
                return None
            if isinstance(rho, (int, float)):
                return (float(rho) * torch.eye(d, dtype=self.dtype, device=self.device)).reshape(1, d, d)
            r = self._t(rho)
            if r.ndim == 2:
                return r.reshape(1, d, d)
            if r.ndim == 3 and bool((r == r[:1]).all()):
                # given per step but the same at every step (compute_Rr_Qr's tiling of a matrix, isls/base.py:55-79): kept as
                # one time-invariant block -- the kernels then hold the rows in registers instead of loading them per step,
                # and the feed-forward pass takes its one-hand-off form
                return r[:1].contiguous()
            return r                                   # [N,d,d] or [B,N,d,d]

        has_x, has_u = x_box is not None or x_sets is not None, u_box is not None or u_sets is not None
        self.Qr = weights(rho_x, n) if has_x else None
        # a state weight that is the same at every step but the LAST (a terminal constraint: the arm notebook's bound on the
        # final end-effector position) reaches the record feed-forward passes as one block plus the terminal block
        # (isls_ff_args.Qr_term): they then run the one-hand-off kernel instead of loading a weight row per step
        self.Qr_ff, self.Qr_term = None, None
        if (self.Qr is not None and self.Qr.ndim == 3 and self.Qr.shape[0] == N and N > 2 and bool((self.Qr[:-1] == self.Qr[:1]).all())
                and os.environ.get("ISLS_FF_QR_TERM", "1") != "0"):
            self.Qr_ff, self.Qr_term = self.Qr[:1].contiguous(), self.Qr[-1].contiguous()
        self.Rr = weights(rho_u, m) if has_u else None
        if has_x and self.Qr is None:
            raise ValueError("project_x needs rho_x")
        if has_u and self.Rr is None:
            raise ValueError("project_u needs rho_u")
        # (dx*dx)@Qr precedence (isls/isls.py:473,476): the AL weights are the row sums
        self.wq = None if self.Qr is None else self.Qr.sum(-1).contiguous()
        self.wr = None if self.Rr is None else self.Rr.sum(-1).contiguous()

        def bounds(box, d):
            if box is None:
                return None, None
            lo, hi = box
            lo = self._t(lo) if not isinstance(lo, (int, float)) else torch.full((1, d), float(lo), dtype=self.dtype, device=self.device)
            hi = self._t(hi) if not isinstance(hi, (int, float)) else torch.full((1, d), float(hi), dtype=self.dtype, device=self.device)
            if lo.ndim == 1 and lo.numel() == N * d:      # flat [N*d] vectors as the reference's callbacks see them
                lo, hi = lo.reshape(N, d), hi.reshape(N, d)
            return lo, hi

        self.x_lo, self.x_hi = bounds(x_box, n)
        self.u_lo, self.u_hi = bounds(u_box, m)
        self.zx, self.lx = (z(B, N, n), z(B, N, n)) if has_x else (None, None)
        self.zu, self.lu = (z(B, N, m), z(B, N, m)) if has_u else (None, None)
        self.relax = float(relax)
        self._outer_args = None
        self._hess_dirty = True

        def device_sets(cs, d):
            """ConvexSets -> (isls_project_args descriptor, first column, scratch) with the operands on the device."""
            if cs is None:
                return None, 0, None
            if cs.dim != d:
                raise ValueError(f"constraint set is defined on rows of dimension {cs.dim}, expected {d}")
            work = z(B, N, d)
            wrap = lambda a: (torch.as_tensor(a, device=self.device) if a.dtype.kind in "iu" else self._t(a))   # noqa: E731
            desc = capi.Kernels.project_args_chain(work, work, cs.stages(), wrap=wrap)
            return desc, cs.cols[0], work

        self.x_sets, self.x_col0, self.x_work = device_sets(x_sets, n)
        self.u_sets, self.u_col0, self.u_work = device_sets(u_sets, m)
        if x_sets is not None:
            self.x_lo = self.x_hi = None
        if u_sets is not None:
            self.u_lo = self.u_hi = None

    # ---- single kernels -----------------------------------------------------------------------------------
    def evaluate_cost(self, out=None):
        """cost of the nominal into `out` (default: self.cost); also refreshes c0x, c0u about it"""
        self.kern.expand_quadratic(self.Qtab, self.ztab, self.seq, self.u_std, self.c0x, self.c0u,
                                   xhat=self.xhat, uhat=self.uhat, cost=self.cost if out is None else out, cost_model=self.cost_model,
                                   cost_par=self.cost_par, q_nonzero=self.q_nonzero, stream=_stream_ptr())

    def linearize(self):
        # a state-independent model (double integrator, dense LTI) has ONE linearisation: it is written for every trajectory,
        # active or not, and advance() then leaves A, Bm alone for as long as they stay the model's
        static = self.model in (capi.MODEL_DI, capi.MODEL_LTI)
        self.kern.linearize(self.model, self.model_par, self.xhat, self.uhat, self.A, self.Bm,
                            active=None if static else self.outer_active, stream=_stream_ptr())
        self._mark_ab_made()
        self._ab_static = self._ab_made if static else None

    # ---- whose A, Bm the buffers hold: the feed-forward passes may use the model's structure only for the model's own
    # linearisation (isls_ff_args.lin_on), never for a caller's A, B
    def _mark_ab_made(self):
        self._ab_made = (self.model, self.A.data_ptr(), self.Bm.data_ptr())
        self._ab_caller = False

    def ab_from_caller(self):
        """A, Bm were written by somebody else (AB setter, get_AB callback): the dense records are the only valid form.
        A cached isls_outer_args block keeps its pointers; run_outer() rewrites its hint fields when this state has changed."""
        self._ab_made = None
        self._ab_static = None
        self._ab_caller = True

    def _ab_is_static(self):
        """A, Bm hold the one linearisation of a state-independent model, written for every trajectory"""
        st = getattr(self, "_ab_static", None)
        return st is not None and st == (self.model, self.A.data_ptr(), self.Bm.data_ptr())

    def _apply_ff_lin(self, blocks, rec, seg=None):
        """(re)write the hint fields of marshalled isls_gain_args / isls_ff_args blocks for the state of A, Bm now"""
        lin = self.ff_lin(rec, seg)
        for a in blocks:
            capi.Kernels._set_lin(a, lin, self.B, self.dtype)
        return lin

    def _structure_expected(self):
        """Will the passes of this engine get the model hint, as far as can be told before A, B are linearised: a model whose
        structure the passes know, no A, B handed in by a caller, the forms not switched off, time-invariant weights."""
        if not self.fast_dims or not getattr(self, "use_model_structure", True) or getattr(self, "_ab_caller", False):
            return False
        if os.environ.get("ISLS_FF_LEAN", "1") == "0" or os.environ.get("ISLS_FF_V2", "1") == "0" or os.environ.get("ISLS_FF_RECORD", "1") == "0":
            return False
        if self.model not in (capi.MODEL_DI, capi.MODEL_ARM3R, capi.MODEL_CAR):
            return False
        inv = lambda W: W is None or W.ndim < 3 or W.shape[-3] == 1       # noqa: E731
        Qr_ok = inv(self.Qr) or getattr(self, "Qr_term", None) is not None
        return inv(self.Rr) and Qr_ok

    def ff_lin(self, rec, seg=None):
        """(model id, parameters) for isls_gain_args.lin_on / isls_ff_args.lin_on, or None.  The hint makes the gain pass write
        the LEAN records and the feed-forward passes read them, so it is given only when every pass on these records can take
        the structured form: the packed records are in use, A and Bm are what isls_linearize wrote into the buffers the engine
        holds now for the model set now, the model is one whose structure the passes know, the ADMM weights are the same at
        every step (a terminal block apart), and the passes run sequentially (`seg`: the time-parallel form builds its
        operators from the dense records).  ISLS_FF_LEAN=0 / use_model_structure = False switch the forms off."""
        if rec is None or seg is not None or not self.fast_dims or not getattr(self, "use_model_structure", True):
            return None
        if os.environ.get("ISLS_FF_LEAN", "1") == "0" or os.environ.get("ISLS_FF_V2", "1") == "0":
            return None
        if self.model not in (capi.MODEL_DI, capi.MODEL_ARM3R, capi.MODEL_CAR) or self.model_par is None:
            return None
        if getattr(self, "_ab_made", None) != (self.model, self.A.data_ptr(), self.Bm.data_ptr()):
            return None
        inv = lambda W: W is None or W.ndim < 3 or W.shape[-3] == 1       # noqa: E731
        Qr_ff, _ = self._ff_weights(rec)
        if not (inv(self.Rr) and inv(Qr_ff)):
            return None
        return (self.model, self.model_par)

    def _shared_hessian(self):
        """True when the cost Hessians are the same arrays for every trajectory of the batch: via-point cost with a batch-
        shared Q table and batch-shared ADMM weights (Cxx_t = 2 Q_seq[t] + 2 Qr_t does not depend on the nominal).  They
        are then written once as [N,n,n] / [N,m,m] and handed to the gain pass with a zero batch stride (SURVEY 8(d):
        "drop the Cxx,Cuu terms when they are shared tables"), instead of B identical copies."""
        return (self.allow_shared_hessian and self.B > 1 and self.cost_model == capi.COST_VIA
                and self.Qtab is not None and self.Qtab.ndim == 3
                and (self.Qr is None or self.Qr.ndim <= 3) and (self.Rr is None or self.Rr.ndim <= 3))

    def hessians(self):
        """(Cxx, Cuu) operands of the gain pass: the batch-shared [1,N,.,.] pair when the cost allows it (expand() then
        writes that form), else the per-trajectory arrays."""
        if not self._shared_hessian():
            return self.Cxx, self.Cuu
        if getattr(self, "_Cxx_sh", None) is None:
            z = lambda *sh: torch.zeros(*sh, dtype=self.dtype, device=self.device)   # noqa: E731
            self._Cxx_sh, self._Cuu_sh = z(1, self.N, self.n, self.n), z(1, self.N, self.m, self.m)
            self._c0_sh = (z(1, self.N, self.n), z(1, self.N, self.m))
        return self._Cxx_sh, self._Cuu_sh

    def expand(self, with_hessian=True):
        shared = with_hessian and self._shared_hessian()
        self.kern.expand_quadratic(self.Qtab, self.ztab, self.seq, self.u_std, self.c0x, self.c0u,
                                   xhat=self.xhat, uhat=self.uhat,
                                   Cxx=self.Cxx if with_hessian and not shared else None,
                                   Cuu=self.Cuu if with_hessian and not shared else None,
                                   Qr=self.Qr, Rr=self.Rr, active=self.outer_active, cost_model=self.cost_model,
                                   cost_par=self.cost_par, q_nonzero=self.q_nonzero, stream=_stream_ptr())
        if shared and getattr(self, "_hess_dirty", True):       # constants of the problem: written once per cost / weights
            Cxx, Cuu = self.hessians()
            self._hess_dirty = False
            self.kern.expand_quadratic(self.Qtab, self.ztab[:1] if self.ztab.ndim == 3 else self.ztab, self.seq, self.u_std,
                                       *self._c0_sh, Cxx=Cxx, Cuu=Cuu, Qr=self.Qr, Rr=self.Rr, stream=_stream_ptr())

    def ff_record(self):
        """Packed step records [A + B K | B | K | fac] the gain pass writes for the feed-forward passes (isls_gain_args.rec /
        isls_ff_args.rec), or None when switched off (ISLS_FF_RECORD=0).  Only the drivers that run the gain pass
        themselves right before the feed-forward passes use it: the records are stale once K / fac are replaced."""
        if os.environ.get("ISLS_FF_RECORD", "1") == "0" or not self.fast_dims:
            return None
        if getattr(self, "_ffrec", None) is None:
            self._ffrec = torch.zeros(capi.ff_record_elems(self.B, self.N, self.n, self.m), dtype=self.dtype, device=self.device)
        return self._ffrec

    def gain(self, active=None, rec=None, seg=None, structured=True):
        """Gain pass; with `rec` the caller promises to run its feed-forward passes on the records, and Quu / fac / Qux
        (which only those passes would read) are not written.  `seg`: the segment plan those passes will use (ff_lin);
        `structured=False`: the caller's passes cannot take the model-structured form (it hands them other weights)."""
        full = rec is None
        lin = self.ff_lin(rec, seg) if structured else None
        self._rec_layout = (None if rec is None else rec.data_ptr(), lin is not None)   # which records, lean or dense
        self.kern.riccati_gain(self.A, self.Bm, *self.hessians(), self.K, self.Quu if full else None,
                               self.fac if full else None, self.Qux if full else None,
                               Cux=self.Cux, solve_mode=self.solve_mode, status=self.status, active=active, rec=rec,
                               lin=lin, stream=_stream_ptr())

    def rec_lin(self, rec, seg=None):
        """The hint for a feed-forward pass on `rec` as the last gain pass left it: lean records need the structured form (and
        raise when it does not apply any more), dense ones the dense form."""
        if rec is None:
            return None
        ptr, lean = getattr(self, "_rec_layout", (None, False))
        if ptr != rec.data_ptr() or not lean:
            return None
        lin = self.ff_lin(rec, seg)
        if lin is None:
            raise capi.IslsError("the gain pass wrote the records in the model-structured layout, which this feed-forward pass cannot "
                                 "read (weights, segments or A, B changed since): run the gain pass again")
        return lin

    def _ff_weights(self, rec):
        """(Qr, Qr_term) operands of a feed-forward pass: the terminal-block form on the packed records when it applies"""
        rr_inv = self.Rr is None or self.Rr.ndim < 3 or self.Rr.shape[-3] == 1
        if rec is not None and getattr(self, "Qr_term", None) is not None and rr_inv and os.environ.get("ISLS_FF_V2", "1") != "0":
            return self.Qr_ff, self.Qr_term
        return self.Qr, None

    def feedforward(self, active=None, seg=None, rec=None):
        Qr, Qr_term = self._ff_weights(rec)
        self.kern.riccati_ff(self.A, self.Bm, self.c0x, self.c0u, self.K, self.Quu, self.fac, self.Qux, self.k,
                             Qr=Qr, Qr_term=Qr_term, Rr=self.Rr, xhat=self.xhat, uhat=self.uhat, zx=self.zx, lx=self.lx,
                             zu=self.zu, lu=self.lu, solve_mode=self.solve_mode, active=active, seg=seg, rec=rec,
                             lin=self.rec_lin(rec, seg), stream=_stream_ptr())

    def rollout(self, L, flags=0, cost_all=None, active=None):
        self.kern.rollout_ls(self.model, self.model_par, self.K, self.k, self.xhat, self.uhat, self.alphas[:L],
                             self.Qtab, self.ztab, self.seq, self.u_std, self.xx, self.xu, best=self.best,
                             cost_new=self.cost_new, cost_all=cost_all, wq=self.wq, wr=self.wr, zx=self.zx,
                             lx=self.lx, zu=self.zu, lu=self.lu, cost_cur=self.cost, flags=flags,
                             status=self.status, active=active, q_nonzero=self.q_nonzero, cost_model=self.cost_model,
                             cost_par=self.cost_par, stream=_stream_ptr())

    def admm_update(self, tol_abs, tol_rel, active=None):
        self.kern.admm_update(self.xx, self.xu, self.res, zx=self.zx, lx=self.lx, zu=self.zu, lu=self.lu,
                              x_lo=self.x_lo, x_hi=self.x_hi, u_lo=self.u_lo, u_hi=self.u_hi, relax=self.relax,
                              tol_abs=tol_abs, tol_rel=tol_rel, res_prev=self.res_prev, active=active,
                              iters=self.admm_iters, x_sets=self.x_sets, x_col0=self.x_col0, x_work=self.x_work,
                              u_sets=self.u_sets, u_col0=self.u_col0, u_work=self.u_work, stream=_stream_ptr())

    # ---- time-parallel feed-forward pass (isls_ffseg): operators from the gain pass, reused by J ADMM iterations
    def ff_seg(self, nseg_requested=None):
        """Segment descriptor over engine-owned buffers, or None for the sequential recursion."""
        if not self.fast_dims:
            return None                                        # the generic kernels recurse sequentially
        if nseg_requested is None:
            # measured on MI355X: from ~2k trajectories on the pass is bound by HBM throughput whatever its
            # shape (82-87 us for 1, 2 or 3 segments at B=4096; 45 / 36+29 / 39+26 us pass+prepare at B=2048), so the
            # sequential recursion wins: no operators to prepare per gain pass, no stitch launch per ADMM iteration.
            # Smaller batches are bound by the N dependent steps and keep the time-parallel form (B=512: 42 us sequential,
            # 30 us in four segments).  Where the model-structured passes apply (ff_lin) they exist for the sequential
            # recursion only and carry the cheaper gain pass with them: outer iteration at n=6, m=3 (tools/kbench.py, one box)
            # B=256: 504 us segmented / 511 us structured, 512: 517 / 510, 1024: 553 / 535, 1536: 608 / 559 -- sequential from 512.
            seq_from = 512 if self._structure_expected() else 2048
            nseg_requested = int(os.environ.get("ISLS_FF_NSEG", "1" if self.B >= seq_from else "4"))
        nseg, seg_len = self.kern.ff_segments(self.N, nseg_requested)
        if nseg < 2:
            return None
        if getattr(self, "_seg_bufs", None) is None or self._seg_bufs[1].shape[1] != nseg:
            z = lambda *shape: torch.zeros(*shape, dtype=self.dtype, device=self.device)
            self._seg_bufs = (z(self.B, self.N, self.m, self.n), z(self.B, nseg, self.n, self.n), z(self.B, nseg, self.n))
        return capi.Kernels.ff_seg(*self._seg_bufs, seg_len)

    def feedforward_prepare(self, seg, active=None, rec=None):
        self.kern.riccati_ff_prepare(self.A, self.Bm, self.K, self.Quu, self.fac, self.Qux, seg,
                                     solve_mode=self.solve_mode, active=active, rec=rec, stream=_stream_ptr())

    # ---- one outer iteration, enqueued by the C driver in one call --------------------------------------------
    def build_outer(self, L, J, tol_abs=0.0, tol_rel=0.0, log=None, ff_nseg=None, begin_done=False):
        """Marshal the argument block of isls_ilqr_admm_outer once; it stays valid while buffers are not re-allocated.
        begin_done: the caller ends every outer iteration with `advance()`, which also makes the ADMM restart of the next one."""
        K = capi.Kernels
        rec = self.ff_record()
        seg = self.ff_seg(ff_nseg)
        # with the records, nothing in this driver reads Quu / fac / Qux: the gain pass then skips those stores
        full = rec is None
        gain = K.gain_args(self.A, self.Bm, *self.hessians(), self.K, self.Quu if full else None, self.fac if full else None,
                           self.Qux if full else None, Cux=self.Cux, solve_mode=self.solve_mode, status=self.status,
                           active=self.admm_active, rec=rec, lin=self.ff_lin(rec, seg))
        Qr_ff, Qr_term = self._ff_weights(rec)
        ff = K.ff_args(self.A, self.Bm, self.c0x, self.c0u, self.K, self.Quu, self.fac, self.Qux, self.k,
                       Qr=Qr_ff, Qr_term=Qr_term, Rr=self.Rr, xhat=self.xhat, uhat=self.uhat, zx=self.zx, lx=self.lx, zu=self.zu,
                       lu=self.lu, solve_mode=self.solve_mode, active=self.admm_active, seg=seg, rec=rec,
                       lin=self.ff_lin(rec, seg))
        ro = K.rollout_args(self.model, self.model_par, self.K, self.k, self.xhat, self.uhat, self.alphas[:L],
                            self.Qtab, self.ztab, self.seq, self.u_std, self.xx, self.xu, best=self.best,
                            cost_new=self.cost_new, wq=self.wq, wr=self.wr, zx=self.zx, lx=self.lx, zu=self.zu,
                            lu=self.lu, cost_cur=self.cost, flags=0, status=self.status, active=self.admm_active,
                            q_nonzero=self.q_nonzero, cost_model=self.cost_model, cost_par=self.cost_par)
        admm = K.admm_args(self.xx, self.xu, self.res, zx=self.zx, lx=self.lx, zu=self.zu, lu=self.lu,
                           x_lo=self.x_lo, x_hi=self.x_hi, u_lo=self.u_lo, u_hi=self.u_hi, relax=self.relax,
                           tol_abs=tol_abs, tol_rel=tol_rel, res_prev=self.res_prev, active=self.admm_active,
                           iters=self.admm_iters, x_sets=self.x_sets, x_col0=self.x_col0, x_work=self.x_work,
                           u_sets=self.u_sets, u_col0=self.u_col0, u_work=self.u_work)
        self._outer_args = capi.OuterArgs(gain=gain, ff=ff, ro=ro, admm=admm, J=int(J), skip_gain=0, begin_done=int(bool(begin_done)))
        self._outer_rec, self._outer_seg = rec, seg
        self._outer_lin_state = (getattr(self, "_ab_made", None), getattr(self, "use_model_structure", True))
        self._advance_args = None
        self._outer_args.log = capi._ptr(log)
        self._outer_args.outer_active = capi._ptr(self.outer_active)
        self._outer_log = log
        return self._outer_args

    def run_outer(self):
        """gain -> J x [ff -> rollout/line-search -> ADMM update] on the current stream (no host sync)."""
        fn = getattr(library(), f"isls_ilqr_admm_outer_{self.sfx}")
        fn.restype = ctypes.c_int
        state = (getattr(self, "_ab_made", None), getattr(self, "use_model_structure", True))
        if state != getattr(self, "_outer_lin_state", ()):     # A, Bm changed hands since the block was marshalled / last run
            self._apply_ff_lin((self._outer_args.gain, self._outer_args.ff), self._outer_rec, self._outer_seg)
            self._outer_lin_state = state
        if self._outer_rec is not None:                        # the driver's gain pass leaves the records in this layout
            self._rec_layout = (self._outer_rec.data_ptr(), bool(self._outer_args.gain.lin_on))
        rc = fn(ctypes.byref(self._outer_args), ctypes.c_void_p(_stream_ptr()))
        if rc != capi.OK:
            raise capi.IslsError(f"isls_ilqr_admm_outer_{self.sfx} -> {rc}")

    def accept_x_step(self, tol_cost=-1.0, tol_osc=-1.0):
        """nominal_values <- last x-step of the ADMM (isls/isls.py:488), cost_log tail and the outer stop
        rules (isls/isls.py:493-499) for the trajectories still iterating; all on the device."""
        self.kern.accept_step(self.xx, self.xu, self.cost_new, self.xhat, self.uhat, self.cost,
                              cost_hist=self.cost_hist, hist_len=self.hist_len, tol_cost=tol_cost, tol_osc=tol_osc,
                              outer_active=self.outer_active, stream=_stream_ptr())

    def begin_outer(self):
        """The ADMM restart at the start of an outer iteration (isls/isls.py:414-415,482; admm.py:25-26) for a driver built with
        begin_done=True: made once before the first iteration (set-up path, torch ops); `advance()` makes the later ones."""
        act = self.outer_active.to(torch.bool)
        self.admm_active.copy_(self.outer_active)
        self.admm_iters.masked_fill_(act, 0)
        for lam in (self.lx, self.lu):
            if lam is not None:
                lam.masked_fill_(act.view(-1, 1, 1), 0.0)
        self.res_prev.masked_fill_(act.view(-1, 1), 1e6)

    def advance(self, tol_cost=-1.0, tol_osc=-1.0, linearize=True):
        """End of an outer iteration and start of the next in one launch (isls_outer_advance_*): accept_x_step(), the ADMM
        restart (admm_active, lambda, residual history), then linearize() and expand() about the new nominal for the
        trajectories still iterating.  Pair it with build_outer(..., begin_done=True).  `linearize=False` leaves A, B alone (a
        shared LTI pair).  The cost Hessians must be the batch-shared tables (written once by expand())."""
        K = capi.Kernels
        if linearize and getattr(self, "use_model_structure", True) and self._ab_is_static():
            linearize = False                                  # the model's one linearisation is in place: nothing to rewrite
        key = (float(tol_cost), float(tol_osc), bool(linearize))
        if getattr(self, "_advance_args", None) is None or self._advance_args[0] != key:
            if not self._shared_hessian():
                raise capi.IslsError("Engine.advance() serves the batch-shared cost Hessians; use accept_x_step / linearize / expand")
            if getattr(self, "_hess_dirty", True):
                self.expand()                                  # writes the shared Hessian tables once
            acc = K.accept_args(self.xx, self.xu, self.cost_new, self.xhat, self.uhat, self.cost, cost_hist=self.cost_hist,
                                hist_len=self.hist_len, tol_cost=tol_cost, tol_osc=tol_osc, outer_active=self.outer_active)
            lin = K.linearize_args(self.model, self.model_par, self.xhat, self.uhat, self.A, self.Bm) if linearize else None
            exp = K.expand_args(self.Qtab, self.ztab, self.seq, self.u_std, self.c0x, self.c0u, xhat=self.xhat, uhat=self.uhat,
                                Qr=self.Qr, Rr=self.Rr, cost_model=self.cost_model, cost_par=self.cost_par, q_nonzero=self.q_nonzero)
            self._advance_args = (key, K.advance_args(acc, lin, exp, admm_active=self.admm_active, iters=self.admm_iters,
                                                      lx=self.lx, lu=self.lu, res_prev=self.res_prev))
        self.kern.outer_advance(self._advance_args[1], self.sfx, stream=_stream_ptr())
        if linearize:
            self._mark_ab_made()                               # the launch linearised the trajectories still iterating

    def reduce(self, table=None, rank=0):
        """[sum cost, max prim, max dual, #active, #failed] of the local shard, left on the device: in `out5`, or straight
        in row `rank` of the all-reduce's [W,5] `table` (its other rows zeroed) -- one launch either way."""
        if table is not None:
            self.kern.reduce_convergence_table(self.cost, self.res, self.outer_active, self.status, table, rank,
                                               stream=_stream_ptr())
            return table
        self.kern.reduce_convergence(self.cost, self.res, self.outer_active, self.status, self.out5, stream=_stream_ptr())
        return self.out5
