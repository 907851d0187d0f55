"""Feedback columns [d, phi] in DP form: batched `iSLS.isls_admm` (isls/isls.py:503-712) and the state-constrained route of
`SLS.ADMM_SLS` (isls/sls.py:319-454).

The reference solves one dense (N m)^2 system for the 1 + dim columns per x-step and multiplies by the dense transfer
matrices.  Here every column is the minimiser of a time-varying LQ problem about the nominal (see `isls_columns_args` in
include/isls_hip.h), so the x-step of an ADMM iteration is

    C feed-forward passes (isls_riccati_ff)  ->  isls_columns_rollout   [-> open-loop line search on column 0, isls_admm only:
    isls_rollout_ls with zero gains = `rollout_batch(x_nom, u_nom + alpha d_u)`, isls.py:593-606]

after ONE Riccati gain pass, and the z-step is isls_columns_admm around the row projection (isls_project_rows for
`projections.ConvexSets`, the caller's numpy function otherwise).  Results equal the dense form up to rounding, including
the last control (SURVEY 8a quirk i), which the dense form sets from its own cost term.
"""
import os

import numpy as np
import torch

from . import _capi as capi
from .engine import _stream_ptr
from .projections import ConvexSets


def _row_projection(project, C):
    """project_x / project_u -> None, ConvexSets on rows of dimension C (device) or a callable (host)."""
    if project is False or project is None:
        return None
    if isinstance(project, ConvexSets):
        if project.dim != C or project.cols != (0, C):
            raise ValueError(f"the projection acts on rows [d, phi] of dimension {C}; this ConvexSets acts on "
                             f"columns {project.cols} of rows of dimension {project.dim}")
        return project
    if callable(project):
        return project
    raise TypeError("project_x / project_u must be False, a projections.ConvexSets or a callable on the rows")


class ColumnSolver:
    """Device state and steps of the ADMM over the feedback columns of one engine (gain pass once, then x-steps and z-steps)."""

    def __init__(self, engine, C, project_x, project_u, nominal_in_projection):
        e = self.e = engine
        self.C, self.nominal_in_projection = int(C), nominal_in_projection
        B, N, n, m = e.B, e.N, e.n, e.m
        z = lambda *s: torch.zeros(*s, dtype=e.dtype, device=e.device)        # noqa: E731
        self.kcol, self.dx, self.du = z(C, B, N, m), z(C, B, N, n), z(C, B, N, m)
        self.zero_x, self.zero_u = z(1, 1, n), z(1, 1, m)
        self.blocks = {}
        for key, proj, d in (("x", _row_projection(project_x, C), n), ("u", _row_projection(project_u, C), m)):
            if proj is None:
                self.blocks[key] = None
                continue
            blk = dict(xx=self.dx if key == "x" else self.du, z=z(C, B, N, d), l=z(C, B, N, d), z_prev=z(C, B, N, d),
                       work=z(B, N * d, C), proj=proj, desc=None)
            if isinstance(proj, ConvexSets):
                wrap = lambda a: (torch.as_tensor(a, device=e.device) if a.dtype.kind in "iu" else e._t(a))   # noqa: E731
                blk["desc"] = capi.Kernels.project_args_chain(blk["work"], blk["work"], proj.stages(), wrap=wrap, active=e.admm_active)
            self.blocks[key] = blk
        self.constrained = self.blocks["x"] is not None or self.blocks["u"] is not None

    def prepare(self, active):
        """gain pass (+ packed records, + operators of the time-parallel feed-forward passes) for the current expansion"""
        e = self.e
        for key, W in (("x", e.Qr), ("u", e.Rr)):                              # residual weights of the projected blocks
            if self.blocks[key] is not None:
                if W is None:
                    raise ValueError(f"project_{key} needs rho_{key}")
                self.blocks[key]["W"] = W
        self.rec = e.ff_record()
        self.seg = e.ff_seg()                                   # planned before the gain pass: it decides the records' layout
        inv = lambda W: W is None or W.ndim < 3 or W.shape[-3] == 1                      # noqa: E731
        structured = inv(e.Qr) and inv(e.Rr)                    # x_step hands e.Qr, e.Rr over as they are
        if (self.seg is not None and structured and self.C * e.B >= 2048 and "ISLS_FF_NSEG" not in os.environ
                and e.ff_lin(self.rec, None) is not None):
            # the C columns of B problems stream like a batch of C B trajectories, and the model-structured form exists for the
            # sequential recursion only (the segment operators come from the dense records): at 4 x 1024 arm columns the
            # structured sequential pass takes 83 us, the dense one in four segments 107 us
            self.seg = None
        e.gain(active=active, rec=self.rec, seg=self.seg, structured=structured)
        self.seg_cols = None
        if self.seg is not None:
            e.feedforward_prepare(self.seg, active=active, rec=self.rec)
            # the same operators with a segment-start scratch per column ([C,B,nseg,n]) for the one-launch form
            G, Psi, v = e._seg_bufs
            if getattr(self, "_vcols", None) is None or self._vcols.shape[2] != v.shape[1]:
                self._vcols = torch.zeros(self.C, *v.shape, dtype=e.dtype, device=e.device)
            self.seg_cols = capi.Kernels.ff_seg(G, Psi, self._vcols.view(self.C * v.shape[0], *v.shape[1:])[:v.shape[0]], self.seg.seg_len)
            self.seg_cols.v = self._vcols.data_ptr()
        self.Cuu = e.hessians()[1]

    def restart(self, active):
        """lmb <- 0, residual history <- 1e6, every active problem iterates (z keeps its value: warm start)"""
        e = self.e
        e.admm_active.copy_(active)
        e.admm_iters.zero_()
        e.res_prev.fill_(1e6)
        for blk in self.blocks.values():
            if blk is not None:
                blk["l"].zero_()

    def _columns_in_one_launch(self):
        """The C passes ride in one launch on the packed records when the ADMM weights do not depend on the time step (the
        one-hand-off kernel's condition); ISLS_ADMM_FF_COLUMNS=0 keeps one launch per column."""
        e = self.e
        inv = lambda W: W is None or W.ndim < 3 or W.shape[-3] == 1                      # noqa: E731
        return (self.rec is not None and inv(e.Qr) and inv(e.Rr) and os.environ.get("ISLS_ADMM_FF_COLUMNS", "1") != "0"
                and os.environ.get("ISLS_FF_V2", "1") != "0")

    def x_step(self):
        """[d_x, phi_x], [d_u, phi_u] for the targets z - lmb: C feed-forward passes, then the column rollout.  A weight that
        is set without a projected block pulls towards zero (its target is the zero vector), as in the reference, where
        Qr / Rr enter the normal equations whether or not the block is projected (isls.py:568-573, sls.py:342-349)."""
        e, bx, bu = self.e, self.blocks["x"], self.blocks["u"]
        act = e.admm_active
        if not hasattr(self, "_zero_zx"):
            zz = lambda d: torch.zeros(e.B, e.N, d, dtype=e.dtype, device=e.device)       # noqa: E731
            self._zero_zx, self._zero_zu = zz(e.n), zz(e.m)
            zc = lambda d: torch.zeros(self.C, e.B, e.N, d, dtype=e.dtype, device=e.device)   # noqa: E731
            self._zero_cx, self._zero_cu = (zc(e.n) if bx is None else None), (zc(e.m) if bu is None else None)
        ff_kw = dict(Qr=e.Qr, Rr=e.Rr, solve_mode=e.solve_mode, active=act, rec=self.rec, lin=e.rec_lin(self.rec, self.seg), stream=_stream_ptr())
        if self._columns_in_one_launch():
            # all C feed-forward passes as ONE launch (isls_ff_args ncol): the columns share the packed records, each has its
            # own targets z - lmb and its own k; the cost gradients act on column 0
            zx, lx = (bx["z"], bx["l"]) if bx is not None else (self._zero_cx, self._zero_cx)
            zu, lu = (bu["z"], bu["l"]) if bu is not None else (self._zero_cu, self._zero_cu)
            with e.timed("riccati_ff"):
                e.kern.riccati_ff(e.A, e.Bm, e.c0x, e.c0u, e.K, e.Quu, e.fac, e.Qux, self.kcol,
                                  zx=zx if e.Qr is not None else None, lx=lx if e.Qr is not None else None,
                                  zu=zu if e.Rr is not None else None, lu=lu if e.Rr is not None else None,
                                  seg=self.seg_cols, ncol=self.C, **ff_kw)
        else:
            for c in range(self.C):
                zx, lx = (bx["z"][c], bx["l"][c]) if bx is not None else (self._zero_zx, self._zero_zx)
                zu, lu = (bu["z"][c], bu["l"][c]) if bu is not None else (self._zero_zu, self._zero_zu)
                with e.timed("riccati_ff"):
                    e.kern.riccati_ff(e.A, e.Bm, e.c0x if c == 0 else self.zero_x, e.c0u if c == 0 else self.zero_u, e.K, e.Quu,
                                      e.fac, e.Qux, self.kcol[c], zx=zx if e.Qr is not None else None,
                                      lx=lx if e.Qr is not None else None, zu=zu if e.Rr is not None else None,
                                      lu=lu if e.Rr is not None else None, seg=self.seg, **ff_kw)
        with e.timed("columns_rollout"):
            e.kern.columns_rollout(e.A, e.Bm, self.Cuu, e.c0u, e.K, self.kcol, self.dx, self.du,
                                   Rr=e.Rr if bu is not None else None, zu=bu["z"] if bu is not None else None,
                                   lu=bu["l"] if bu is not None else None, active=act, stream=_stream_ptr())

    def iteration_args(self, L, zero_K, relax, tol_abs, tol_rel):
        """isls_columns_iteration_args of one ADMM iteration on this solver's buffers (call after prepare()): feed-forward passes
        of the C columns, column rollout, open-loop line search over alphas[:L] on column 0 (L = 0: none), z-step."""
        e, bx, bu, K = self.e, self.blocks["x"], self.blocks["u"], capi.Kernels
        C_ = self.C
        zc = lambda d: torch.zeros(C_, e.B, e.N, d, dtype=e.dtype, device=e.device)      # noqa: E731
        if not hasattr(self, "_zero_cx"):
            self._zero_cx, self._zero_cu = (zc(e.n) if bx is None else None), (zc(e.m) if bu is None else None)
        zx, lx = (bx["z"], bx["l"]) if bx is not None else (self._zero_cx, self._zero_cx)
        zu, lu = (bu["z"], bu["l"]) if bu is not None else (self._zero_cu, self._zero_cu)
        ff = K.ff_args(e.A, e.Bm, e.c0x, e.c0u, e.K, e.Quu, e.fac, e.Qux, self.kcol, Qr=e.Qr, Rr=e.Rr,
                       zx=zx if e.Qr is not None else None, lx=lx if e.Qr is not None else None,
                       zu=zu if e.Rr is not None else None, lu=lu if e.Rr is not None else None, solve_mode=e.solve_mode,
                       active=e.admm_active, seg=self.seg_cols, rec=self.rec, ncol=C_, lin=e.rec_lin(self.rec, self.seg))
        if self.rec is None:                                    # array form: the factors of the gain pass
            ff.Quu, ff.fac, ff.Qux = e.Quu.data_ptr(), e.fac.data_ptr(), e.Qux.data_ptr()
        if not self._columns_in_one_launch():
            ff._pad = 1                                         # one feed-forward launch per column
        cols = K.columns_args(e.A, e.Bm, self.Cuu, e.c0u, e.K, self.kcol, self.dx, self.du, Rr=e.Rr if bu is not None else None,
                              zu=bu["z"] if bu is not None else None, lu=bu["l"] if bu is not None else None, active=e.admm_active)
        ls = None
        if L > 0:
            ls = K.rollout_args(e.model, e.model_par, zero_K, self.du[0], e.xhat, e.uhat, e.alphas[:L], e.Qtab, e.ztab, e.seq, e.u_std,
                                e.xx, e.xu, best=e.best, cost_new=e.cost_new, flags=0, status=e.status, active=e.admm_active,
                                q_nonzero=e.q_nonzero, cost_model=e.cost_model, cost_par=e.cost_par)
        admm = None
        if self.constrained:
            noms = {"x": e.xhat, "u": e.uhat}
            for key, blk in self.blocks.items():
                if blk is not None:
                    blk["nom"] = noms[key] if (self.nominal_in_projection and blk["desc"] is not None) else None
            admm = K.columns_admm_args(0, (e.B, e.N, e.n, e.m, C_), e.res, e.res_prev, x=bx, u=bu, relax=relax, tol_abs=tol_abs,
                                       tol_rel=tol_rel, active=e.admm_active, iters=e.admm_iters)
        self._keep_alphas = e.alphas[:L]
        return K.columns_iteration_args(ff, cols, ls, admm, proj_x=bx["desc"] if bx is not None else None,
                                        proj_u=bu["desc"] if bu is not None else None, zero_x=self.zero_x, zero_u=self.zero_u)

    def z_step(self, relax, tol_abs, tol_rel, log_row=None):
        """z = Proj(relax x + (1 - relax) z + lmb), lmb += x - z, rho-scaled residuals and the two stop rules"""
        e, bx, bu = self.e, self.blocks["x"], self.blocks["u"]
        act = e.admm_active
        dims = (e.B, e.N, e.n, e.m, self.C)
        noms = {"x": e.xhat, "u": e.uhat}
        for key, blk in self.blocks.items():
            if blk is not None:
                blk["nom"] = noms[key] if (self.nominal_in_projection and blk["desc"] is not None) else None
        e.kern.columns_admm(0, dims, e.res, e.res_prev, x=bx, u=bu, relax=relax, active=act, stream=_stream_ptr())
        for key, blk in self.blocks.items():
            if blk is None:
                continue
            if blk["desc"] is not None:
                with e.timed("project_rows"):
                    e.kern._call("project_rows", e.sfx, blk["desc"], _stream_ptr())
            else:                                                               # the caller's numpy projection, problem by problem
                rows, on_h = blk["work"].cpu().numpy(), act.cpu().numpy()
                nom_h = noms[key].cpu().numpy() if self.nominal_in_projection else None
                for b in range(e.B):
                    if on_h[b]:
                        arg = rows[b].astype(np.float64)
                        out = blk["proj"](arg, nom_h[b]) if nom_h is not None else blk["proj"](arg)
                        rows[b] = np.asarray(out, dtype=np.float64).reshape(rows[b].shape)
                blk["work"].copy_(e._t(rows))
        e.kern.columns_admm(1, dims, e.res, e.res_prev, x=bx, u=bu, relax=relax, tol_abs=tol_abs, tol_rel=tol_rel, active=act,
                            iters=e.admm_iters, stream=_stream_ptr())
        if log_row is not None:
            log_row.copy_(e.res)

    @staticmethod
    def rows_to_host(cols):
        """[C, B, N, d] columns on the device -> [B, N d, C] float64 on the host (the reference's row layout)"""
        C, B, N, d = cols.shape
        return cols.permute(1, 2, 3, 0).reshape(B, N * d, C).cpu().numpy().astype(np.float64, copy=False)

    def columns(self):
        """(x_x [B, N n, C], x_u [B, N m, C]) of the last x-step in the reference's row layout"""
        return self.rows_to_host(self.dx), self.rows_to_host(self.du)


class _LaggedAny:
    """`bool(mask.any())` without a device sync: the flag of every iteration is copied to pinned host memory behind the
    iteration's kernels, and the loop only LOOKS at flags whose copy has already landed (`event.query()`), so the host keeps
    queueing iterations ahead of the GPU instead of waiting for each one (a device sync per ADMM iteration made `isls_admm`
    host-bound: 1.19 ms per iteration for 0.65 ms of kernels).  The loop may therefore run a few iterations more than the
    reference's `break` -- with every problem inactive the kernels of those iterations touch nothing; `dead_rows()` tells
    afterwards which iterations they were."""

    def __init__(self, device, n):
        self.host = torch.ones(max(1, n), dtype=torch.int32).pin_memory()
        self.dev = torch.ones(max(1, n), dtype=torch.int32, device=device)
        self.ev = []

    def reset(self):
        """start a new sequence of flags on the same buffers (the previous sequence's copies have been queued before)"""
        self.ev = []

    def push(self, mask):
        i = len(self.ev)
        self.dev[i] = mask.any()
        self.host[i:i + 1].copy_(self.dev[i:i + 1], non_blocking=True)
        e = torch.cuda.Event()
        e.record()
        self.ev.append(e)

    def push_flag(self, flag):
        """the same with a one-word device flag somebody else has computed (isls_columns_iteration_args.any_active)"""
        i = len(self.ev)
        self.host[i:i + 1].copy_(flag, non_blocking=True)
        e = torch.cuda.Event()
        e.record()
        self.ev.append(e)

    def seen_all_inactive(self):
        """True when a flag that has already arrived says that no problem was active after its iteration."""
        for i, e in enumerate(self.ev):
            if not e.query():
                break
            if int(self.host[i]) == 0:
                return True
        return False

    def dead_rows(self):
        """Iterations that ran although an earlier one had left no problem active (their log rows are not the reference's)."""
        if not self.ev:
            return []
        self.ev[-1].synchronize()
        first = next((i for i in range(len(self.ev)) if int(self.host[i]) == 0), None)
        return [] if first is None else list(range(first + 1, len(self.ev)))


def isls_admm(self, dim, get_AB=None, get_Cs=None, project_x=False, project_u=False, max_admm_iter=20, k_max=20,
              max_line_search=20, rho_x=None, rho_u=None, alpha=1, threshold=1e-3, verbose=False, log=False):
    """Returns (du [N m], phi_u [N m, dim]) of the last ADMM x-step (with a leading batch axis when batch > 1).

    project_x / project_u: `projections.ConvexSets` over the rows [nominal + d, phi] (the shift by the nominal of
    notebook cell 25 is applied on the device), or a callable `(rows [N d, 1 + dim], nominal [N, d]) -> rows` in the
    reference's convention (host round trip per ADMM iteration).  `self.admm_iters` holds the executed ADMM iterations of
    the last outer iteration per problem, `self.admm_logs` their (prim, dual) residuals [J, B, 2]."""
    self._check_get_Cs(get_Cs)                                                  # a callable cost needs its get_Cs
    e = self.engine
    B, N, n, m, C = self.batch, self.N, self.x_dim, self.u_dim, int(dim) + 1
    if not 1 <= dim <= n or C > capi.MAX_ROW_DIM:
        raise ValueError(f"dim must be in [1, {min(n, capi.MAX_ROW_DIM - 1)}]")
    cs = ColumnSolver(e, C, project_x, project_u, nominal_in_projection=True)
    free = (-np.inf, np.inf)
    e.set_admm(rho_x=rho_x if cs.blocks["x"] is not None else None, rho_u=rho_u if cs.blocks["u"] is not None else None,
               x_box=free if cs.blocks["x"] is not None else None, u_box=free if cs.blocks["u"] is not None else None, relax=alpha)
    dx, du = cs.dx, cs.du
    zero_K = torch.zeros(B, N, m, n, dtype=e.dtype, device=e.device)
    J = int(max_admm_iter) if cs.constrained else 1
    L = int(max_line_search)
    logbuf = torch.zeros(J, B, 2, dtype=e.dtype, device=e.device)
    e.outer_active.fill_(1)
    # Outer-loop state stays on the device: the cost of every outer iteration goes into `costlog` (read once at the end), the
    # two stop rules (isls.py:695-701) run in isls_accept_step on the cost history, and the host looks at `outer_active` only
    # through flags that have already landed in pinned memory -- it never waits for the GPU inside the loop, so the kernels
    # of outer iteration k + 1 are queued while those of k still run (round 2 read status, cost and the active mask back
    # after every outer iteration: 7.4 ms of wall time for 3.6-4.7 ms of kernels).  A batch that went inactive may therefore
    # see an outer iteration or two more; their kernels skip every problem.
    costlog = torch.zeros(int(k_max) + 1, B, dtype=e.dtype, device=e.device)
    costlog[0].copy_(e.cost)
    e.cost_hist.zero_()
    e.cost_hist[:, 0] = e.cost
    e.hist_len.fill_(1)
    mask3 = lambda a: a.to(torch.bool).view(B, 1, 1)                          # noqa: E731
    device_only = not self._host_ls and all(blk is None or blk["desc"] is not None for blk in cs.blocks.values())
    # everything on the device: ONE C call per ADMM iteration (isls_columns_iteration_*: feed-forward passes, column rollout,
    # line search and its step, z-step around the row projections -- ~12 launches on fixed buffers).  Round 2 replayed a HIP
    # graph of the iteration instead; recording it cost 55-80 ms per isls_admm call, more than ten outer iterations take.
    use_driver = device_only and e.profile_events is None and os.environ.get("ISLS_ADMM_DRIVER", "1") != "0"
    drv = dict(args=None, ptrs=None)
    host_sync = self._host_ls or self._host_cost or not device_only or verbose   # host callbacks need the numbers anyway
    outer_count = torch.zeros(B, dtype=torch.int32, device=e.device)
    flags = torch.ones(max(1, J), dtype=torch.int32, device=e.device)          # "a problem is still active" behind every ADMM iteration
    outer_lag = _LaggedAny(e.device, int(k_max))
    inner_lag = _LaggedAny(e.device, J)
    ran = 0

    def captured_pointers():
        """device addresses a recorded ADMM iteration reads or writes through engine attributes a callback may replace"""
        ts = [e.A, e.Bm, e.c0x, e.c0u, e.K, e.Qr, e.Rr, e.xhat, e.uhat, e.xx, e.xu, e.Qtab, e.ztab, e.model_par, e.cost_par,
              cs.rec, cs.Cuu] + (list(e._seg_bufs) if getattr(e, "_seg_bufs", None) is not None and cs.seg is not None else [])
        return tuple(None if t is None else t.data_ptr() for t in ts)

    mark = getattr(self, "_bench_mark", None)                                   # bench.py: t1 behind the first outer iteration (the
    for k in range(k_max):                                                      # call's set-up), t2 behind the loop, before the read-back
        if mark is not None and k == 1:
            import time
            torch.cuda.synchronize()
            mark["t1"] = time.perf_counter()
        outer_count.add_(e.outer_active)
        self._linearize(get_AB)
        self._expand_regularised(get_Cs)                                        # built-in cost on the device, else the caller's get_Cs (isls.py:548-560)
        cs.prepare(e.outer_active)
        cs.restart(e.outer_active)                                              # lmb restarts, z is warm-started (isls.py:613-616)
        inner_lag.reset()
        act = e.admm_active

        def admm_iteration():
            """x-step (C feed-forward passes + column rollout), line search on d_u, z-step: ~25 launches on fixed buffers"""
            cs.x_step()
            # line search on d_u: open-loop rollouts of u_nom + alpha d_u, plain cost, first arg-min (isls.py:593-606)
            if self._host_ls:                                                   # callable model / cost: the open-loop search on the host
                from . import hostpath
                hostpath.line_search(self, L, 0, act, K=np.zeros((B, N, m, n)), k=du[0].cpu().numpy().astype(np.float64), plain_only=True)
            else:
                with e.timed("rollout_ls"):
                    e.kern.rollout_ls(e.model, e.model_par, zero_K, du[0], e.xhat, e.uhat, e.alphas[:L], e.Qtab, e.ztab, e.seq, e.u_std,
                                      e.xx, e.xu, best=e.best, cost_new=e.cost_new, flags=0, status=e.status, active=act,
                                      q_nonzero=e.q_nonzero, cost_model=e.cost_model, cost_par=e.cost_par, stream=_stream_ptr())
            on = mask3(act)
            step = torch.where(act.to(torch.bool), e.alphas[:L][e.best.long()], torch.ones_like(e.cost_new))
            du[0].mul_(step.view(B, 1, 1))                                      # du_opt[:, 0] = alpha* d_u
            dx[0].copy_(torch.where(on, e.xx - e.xhat, dx[0]))                  # dx_opt[:, 0] = x_noms[ind] - x_nom
            if cs.constrained:
                cs.z_step(alpha, threshold, 1e-3)                               # isls.py:626-665

        if use_driver and (drv["args"] is None or drv["ptrs"] != captured_pointers()):
            drv.update(args=cs.iteration_args(L, zero_K, alpha, threshold, 1e-3), ptrs=captured_pointers())
        for j in range(J):
            if use_driver:
                drv["args"].log = logbuf[j].data_ptr() if cs.constrained else None
                drv["args"].any_active = flags[j:j + 1].data_ptr() if cs.constrained else None
                e.kern.columns_iteration(drv["args"], e.sfx, stream=_stream_ptr())
            else:
                admm_iteration()
            if not cs.constrained:
                e.admm_iters.add_(act)
                break
            if use_driver:
                inner_lag.push_flag(flags[j:j + 1])
            else:
                logbuf[j].copy_(e.res)
                inner_lag.push(act)
            if inner_lag.seen_all_inactive():
                break
        # new nominal: x_nom + d_x, u_nom + d_u of the last x-step (isls.py:684-687); the setter evaluates its cost
        oa = e.outer_active.to(e.dtype).view(B, 1, 1)
        e.xhat.addcmul_(dx[0], oa)                                              # x_nom + d_x where the problem still iterates
        e.uhat.addcmul_(du[0], oa)
        # cost log entry of this outer iteration and the two stop rules (|cost - prev| < 1e-4, isls.py:695-697; oscillation of
        # the last eight costs < 1e-3, isls.py:699-701) on the device: isls_accept_step with the nominal as its own x-step;
        # e.cost still holds the cost before this iteration (`prev` of the rule), the new one goes to cost_new
        e.evaluate_cost(out=e.cost_new)
        if self._host_cost:
            self._refresh_host_cost()
            e.cost_new.copy_(e.cost)
            e.cost.copy_(costlog[k])
        ran = k + 1
        costlog[k + 1].copy_(e.cost_new)
        e.kern.accept_step(e.xhat, e.uhat, e.cost_new, e.xhat, e.uhat, e.cost, cost_hist=e.cost_hist, hist_len=e.hist_len,
                           tol_cost=1e-4, tol_osc=1e-3, outer_active=e.outer_active, stream=_stream_ptr())
        outer_lag.push(e.outer_active)
        if host_sync:
            st = e.status.cpu().numpy()
            if (st & capi.ST_NOT_PD).any():
                raise np.linalg.LinAlgError("Quu not positive definite")
            if verbose:
                cost = e.cost.cpu().numpy()
                for b_ in np.nonzero(e.outer_active.cpu().numpy())[0]:
                    print("Iteration number ", k, "iSLS cost: ", cost[b_])
            if not bool(e.outer_active.any().item()):
                break
        elif outer_lag.seen_all_inactive():
            break
    # ---- one synchronisation for the whole call -----------------------------------------------------------------------
    if mark is not None:
        import time
        torch.cuda.synchronize()
        mark["t2"] = time.perf_counter()
    st = e.status.cpu().numpy()
    if (st & capi.ST_NOT_PD).any():
        raise np.linalg.LinAlgError("Quu not positive definite")
    dead = outer_lag.dead_rows()                                                # outer iterations that ran on no problem
    ran = ran - len([d for d in dead if d < ran])
    cl = costlog[:ran + 1].cpu().numpy().astype(np.float64)
    self.outer_iters = outer_count.cpu().numpy().astype(np.int64)              # outer iterations every problem took part in
    for k_ in range(ran):
        self.cost_log.append(cl[k_ + 1] if B > 1 else float(cl[k_ + 1][0]))
    # log rows of ADMM iterations that ran on no problem are not the reference's
    iters_h = e.admm_iters.cpu().numpy()
    for jd in range(int(iters_h.max()) if iters_h.size else 0, J):
        logbuf[jd].zero_()
    self.admm_iters = iters_h
    self.admm_logs = logbuf.cpu().numpy()
    # [d_u, phi_u] is the return value; [d_x, phi_x] (3x the bytes: 29 MB of the 39 MB at B = 1024, n = 9) stays on the
    # device until somebody asks for `_dx_columns` -- reading both back through pageable memory cost 30-40 ms per call
    self._dx_columns_dev, self._dx_columns_host = cs.dx, None
    xu = cs.rows_to_host(cs.du)
    du_out, phi_out = xu[..., 0], xu[..., 1:]
    return (du_out[0], phi_out[0]) if B == 1 else (du_out, phi_out)


def admm_sls_columns(self, project_x, project_u, max_iter, rho_x, rho_u, alpha, tol, PHI_U):
    """`SLS.ADMM_SLS` with state constraints (isls/sls.py:319-454) on the feedback columns: absolute coordinates, columns
    [d, phi] with respect to the initial position (dim = x_dim / 2), no line search (the model is linear), z and lmb start
    at zero, stop rules with the relative tolerance 1e-2 (sls.py:417-430).  project_x / project_u: `ConvexSets` over the
    rows, or the reference's callables `rows -> rows`.  Returns (du, phi_u, logs [iterations, B, 2], iterations [B])."""
    e = self.engine
    B, N, n, m = self.batch, self.N, self.x_dim, self.u_dim
    p = n // 2
    C = p + 1
    cs = ColumnSolver(e, C, project_x, project_u, nominal_in_projection=False)
    # compute_Rr_Qr(dp=False) block-diagonal weights as per-step blocks; they enter the Hessians whether or not the block is
    # projected (sls.py:342-349).  A zero rho is a zero block, not "no weight".
    Qr, Rr = self.compute_Rr_Qr(rho_x=rho_x if rho_x is not None else 0.0, rho_u=rho_u if rho_u is not None else 0.0, dp=True)
    self._set_reg(Qr, Rr, np.zeros(N * n), np.zeros(N * m))
    self._expand_abs()
    e.status.zero_()
    ones = torch.ones(B, dtype=torch.int32, device=e.device)
    cs.prepare(ones)
    if (e.status.cpu().numpy() & capi.ST_NOT_PD).any():
        raise np.linalg.LinAlgError("Singular matrix")
    cs.restart(ones)
    J = int(max_iter)
    logbuf = torch.zeros(J, B, 2, dtype=e.dtype, device=e.device)
    done = 0
    for j in range(J):
        cs.x_step()
        done = j + 1
        if not cs.constrained:                                                  # both residuals are zero: the loop stops at once
            e.admm_iters.fill_(1)
            break
        cs.z_step(alpha, tol, 1e-2, log_row=logbuf[j])
        if not bool(e.admm_active.any().item()):                                # one problem class, a few iterations: the plain test
            break
    xu = cs.rows_to_host(cs.du)
    du = xu[..., 0]
    phi_u = np.concatenate([xu[..., 1:], np.broadcast_to(PHI_U[:, p:], (B,) + PHI_U[:, p:].shape)], axis=-1)
    return du, phi_u, logbuf[:done].cpu().numpy(), e.admm_iters.cpu().numpy()
