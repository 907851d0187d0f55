"""Batched LQT / LQT-ADMM (DP form) front end with the reference's `SLS` class surface.

Reference: `SLS` (isls/sls.py) + `SLSBase` (isls/sls_base.py).  Built on the same HIP kernels as `iSLS`, in
their absolute-coordinate / explicit-inverse mode (`ISLS_SOLVE_INV`, `ISLS_RO_ABSOLUTE`):

    solve_dp(Qr,Rr,ur,xr,return_Qs) -> expand_quadratic + riccati_gain(INV) + riccati_ff   (isls/sls.py:85-166)
    solve_dp_ff                     -> riccati_ff with the cached K, Quu, Quu_inv, Qux       (isls/sls.py:168-202)
    get_trajectory_dp               -> rollout_ls(ABSOLUTE)                                  (isls/sls_base.py:76-89)
    ADMM_LQT_DP                     -> gain once, then per iteration ff + rollout + admm_update (isls/sls.py:298-317)

Config 5 of BASELINE.json (system level synthesis with chance constraints on the controls):

    AB setter -> Sw, Su             transfer matrices, built on the host on first use        (isls/base.py:98-119)
    solve_sls / compute_inverses    dense (N m)^2 set-up, numpy on the host, once per class of problems
                                                                                            (isls/sls.py:205-233, base.py:29-50)
    ADMM_SLS                        the ADMM loop itself runs on the device for all problems of the batch in one launch
                                    (isls_sls_admm: x-step, project_set_convex over the rows, residuals, stop rules)
                                                                                            (isls/sls.py:319-454)
    controller / get_trajectory_sls K = Phi_u Phi_x^-1 (host set-up) and the closed-loop Monte-Carlo rollout (device)
                                                                                            (isls/sls.py:235-242, sls_base.py:91-105)

solve_batch / ADMM_LQT_Batch return the batch form's results through the same Riccati kernels; replanning and the
`*_optimal` helpers are dense host algebra as in the reference.
"""
import numpy as np
import torch

from . import _capi as capi
from . import sls_dense as dense
from .base import Base
from .models import LTI
from .projections import Box, ConvexSets


class SLS(Base):
    def __init__(self, x_dim, u_dim, N, batch=1, dtype=np.float64, device="cuda"):
        super().__init__(x_dim, u_dim, N, batch=batch, dtype=dtype, device=device)
        self.engine.solve_mode = capi.SOLVE_INV
        self.engine.allow_shared_hessian = False               # _expand_abs writes the per-trajectory Cxx / Cuu itself
        self._model = None
        self._Sw = self._Su = None
        self.l_side_invs = None

    # ---- dynamics -------------------------------------------------------------------------------------------
    @property
    def AB(self):
        return [self.A, self.B]

    @AB.setter
    def AB(self, value):
        """LTI dynamics A [n,n], B [n,m] (isls/base.py:98-119; the Sw/Su transfer matrices of the batch form are
        not built: the DP form never needs them)."""
        self.A, self.B = np.asarray(value[0], dtype=np.float64), np.asarray(value[1], dtype=np.float64)
        if self.A.ndim != 2:
            raise NotImplementedError("SLS is the LTI class of the reference (sls.py:139-140); use iSLS for time-varying A,B")
        self._model = LTI(self.A, self.B)
        self._Sw = self._Su = None
        self.l_side_invs = None
        e = self.engine
        e.set_model(self._model.model_id, self._model.params())
        e.A, e.Bm = e._t(self.A).reshape(1, 1, self.x_dim, self.x_dim), e._t(self.B).reshape(1, 1, self.x_dim, self.u_dim)
        e.ab_from_caller()                                      # a shared pair installed as it is: the dense passes, their plan

    def forward_model(self, x, u):
        return self._model(np.asarray(x), np.asarray(u))

    def compute_cost(self, x, u=None, cost_function=None):
        """(x-xd)'Q(x-xd) + u'Ru without the 1/2 (isls/sls_base.py:25-44); x [N,n] / flat, or with leading batch axes."""
        if cost_function is not None:
            return cost_function(x=x, u=u)
        x = np.asarray(x, dtype=np.float64)
        x = x.reshape(x.shape[:-1] + (self.N, self.x_dim)) if x.shape[-1] == self.N * self.x_dim else x
        dx = x - self.zs[self.seq]
        c = np.einsum("...ti,tij,...tj->...", dx, self.Qs[self.seq], dx)
        if u is not None:
            u = np.asarray(u, dtype=np.float64)
            c = c + self.u_std * np.sum(u * u, axis=tuple(range(x.ndim - 2, u.ndim)))
        return float(c) if np.ndim(c) == 0 else c

    # ---- DP solvers -----------------------------------------------------------------------------------------
    def _set_reg(self, Qr, Rr, xr, ur):
        """Regulariser of the ADMM sub-problem (isls/sls.py:104-137): weights Qr [N,n,n] / Rr list, targets xr, ur."""
        e = self.engine
        B, N, n, m = self.batch, self.N, self.x_dim, self.u_dim
        z = lambda *s: torch.zeros(*s, dtype=e.dtype, device=e.device)   # noqa: E731
        e.Qr = None if Qr is None else e._t(np.asarray(Qr))
        e.Rr = None if Rr is None else e._t(np.stack([np.asarray(r) for r in Rr]) if isinstance(Rr, (list, tuple)) else np.asarray(Rr))
        if Qr is not None:
            assert xr is not None
            e.zx, e.lx = e._t(self._batched(np.asarray(xr).reshape(-1, N, n) if np.ndim(xr) > 1 else np.asarray(xr).reshape(N, n), 2)), z(B, N, n)
        else:
            e.zx = e.lx = None
        if Rr is not None:
            assert ur is not None
            e.zu, e.lu = e._t(self._batched(np.asarray(ur).reshape(-1, N, m) if np.ndim(ur) > 1 else np.asarray(ur).reshape(N, m), 2)), z(B, N, m)
        else:
            e.zu = e.lu = None

    def _expand_abs(self):
        e = self.engine
        e.kern.expand_quadratic(e.Qtab, e.ztab, e.seq, e.u_std, e.c0x, e.c0u, Cxx=e.Cxx, Cuu=e.Cuu, Qr=e.Qr, Rr=e.Rr,
                                stream=torch.cuda.current_stream().cuda_stream)

    def _ff_abs(self):
        e = self.engine
        e.kern.riccati_ff(e.A, e.Bm, e.c0x, e.c0u, e.K, e.Quu, e.fac, e.Qux, e.k, Qr=e.Qr, Rr=e.Rr, zx=e.zx, lx=e.lx,
                          zu=e.zu, lu=e.lu, solve_mode=capi.SOLVE_INV, stream=torch.cuda.current_stream().cuda_stream)

    def solve_dp(self, Qr=None, Rr=None, ur=None, xr=None, return_Qs=False):
        assert self.A is not None, "Set the linear dynamics model by self.AB = [A,B] before calling this method."
        assert self.Q is not None, "Set the quadratic cost model by self.set_cost_variables() before calling this method."
        e = self.engine
        self._set_reg(Qr, Rr, xr, ur)
        self._expand_abs()
        e.status.zero_()
        e.gain()
        self._ff_abs()
        if (e.status.cpu().numpy() & capi.ST_NOT_PD).any():
            raise np.linalg.LinAlgError("Singular matrix")
        if return_Qs:
            return self._out(e.K), self._out(e.k), self._out(e.Quu), self._out(e.fac), self._out(e.Qux)
        return self._out(e.K), self._out(e.k)

    def solve_dp_ff(self, K, Quu, Qux, Quu_inv, Qr=None, Rr=None, ur=None, xr=None):
        e = self.engine
        e.K.copy_(e._t(self._batched(K, 3))), e.Quu.copy_(e._t(self._batched(Quu, 3)))
        e.Qux.copy_(e._t(self._batched(Qux, 3))), e.fac.copy_(e._t(self._batched(Quu_inv, 3)))
        self._set_reg(Qr, Rr, xr, ur)
        self._expand_abs()
        self._ff_abs()
        return self._out(e.k)

    def solve(self, x0=None, method='sls'):
        """isls/sls.py:40-58."""
        if method == 'batch':
            assert x0 is not None
            return self.solve_batch(x0)
        if method == 'dp':
            return self.solve_dp()
        return self.solve_sls()

    def get_trajectory_dp(self, x0, K, k, noise_scale=0):
        """u_t = K_t x_t + k_t, x_{t+1} = A x_t + B u_t  (isls/sls_base.py:76-89).  x0 [n] -> one trajectory per problem
        of the batch; x0 [M,n] with batch == 1 evaluates M initial states against the same controller (Monte Carlo)."""
        if noise_scale:
            # process noise comes from numpy's global generator, one draw per step in the reference's order
            # (isls/sls_base.py:76-89): a host loop, so that a seeded run reproduces the reference's trajectories
            from . import hostpath
            K, k = np.asarray(K, dtype=np.float64), np.asarray(k, dtype=np.float64)
            A, Bm = np.asarray(self.A, dtype=np.float64), np.asarray(self.B, dtype=np.float64)
            return hostpath.noisy_closed_loop(lambda x, u: x @ A.T + u @ Bm.T, x0, self.N, self.u_dim,
                                              lambda i, x_log: x_log[:, i] @ K[i].T + k[i], noise_scale)
        x0 = np.asarray(x0, dtype=np.float64)
        e = self.engine
        if x0.ndim == 2 and self.batch == 1 and x0.shape[0] != 1:
            M = x0.shape[0]
            mc = SLS(self.x_dim, self.u_dim, self.N, batch=M, dtype=self.np_dtype, device=e.device)
            mc.AB = [self.A, self.B]
            mc.set_quadratic_cost(self.zs, self.Qs, self.seq, self.u_std)
            return mc.get_trajectory_dp(x0, K, k)
        x0 = self._batched(x0, 1)
        one = torch.ones(1, dtype=e.dtype, device=e.device)
        e.kern.rollout_ls(e.model, e.model_par, e._t(self._batched(K, 3)), e._t(self._batched(k, 2)), e.xhat, e.uhat, one,
                          e.Qtab, e.ztab, e.seq, e.u_std, e.xx, e.xu, x0=e._t(x0), flags=capi.RO_ABSOLUTE,
                          q_nonzero=e.q_nonzero, stream=torch.cuda.current_stream().cuda_stream)
        return self._out(e.xx), self._out(e.xu)

    # ---- LQT-ADMM, DP form --------------------------------------------------------------------------------------
    def ADMM_LQT_DP(self, x0, project_x=False, project_u=False, max_iter=2000, rho_x=None, rho_u=None, alpha=1.,
                    tol=1e-3, verbose=False, log=False):
        """isls/sls.py:298-317: gain pass once, then ADMM iterations of [ff pass, closed-loop rollout from x0,
        z/lambda update] with the stop rules of isls/admm.py:72-85.  Returns (x_flat, u_flat, K, k[, logs])."""
        return self._admm_lqt(x0, project_x, project_u, max_iter, rho_x, rho_u, alpha, tol, log, batch_form=False)

    def ADMM_LQT_Batch(self, x0, project_x=False, project_u=False, max_iter=20, rho_x=None, rho_u=None, alpha=1.,
                       tol=1e-3, verbose=False, log=False):
        """isls/sls.py:252-293 without its dense (N m)^2 algebra: the batch-form x-step
        u = (Su'Q Su + R + Su'Qr Su + Rr)^-1 (...) is the minimiser the Riccati pass computes, except for the last control,
        which never acts on the state and which the dense form sets to (R + Rr)^-1 Rr (z_u - lmb_u) at t = N-1 (SURVEY 8a
        quirk i); z starts at the unconstrained solution (sls.py:268-270).  Returns (x_flat, u_flat[, logs])."""
        out = self._admm_lqt(x0, project_x, project_u, max_iter, rho_x, rho_u, alpha, tol, log, batch_form=True)
        return out if len(out) <= 3 else out[:2] + out[4:]        # the device route also carries K, k (as ADMM_LQT_DP returns them)

    def solve_batch(self, x0):
        """isls/sls.py:60-82: the unconstrained batch-form LQT solution equals the Riccati solution rolled out from x0 (the
        last control is zero in both forms: its only cost term is u'Ru).  Returns (x_opt [N,n], u_opt [N,m])."""
        K, k = self.solve_dp()
        return self.get_trajectory_dp(x0, K, k)

    def _admm_lqt_host(self, x0, px, pu, max_iter, rho_x, rho_u, alpha, tol, log, batch_form):
        """Projections given as arbitrary numpy callables (e.g. the Dykstra closures of the spherical-obstacle notebook):
        the x-step runs on the device, the z-step is the caller's function inside the host mirror of the reference's ADMM()
        (isls/admm.py).  One problem at a time, as the reference; device descriptors (Box, ConvexSets) are the batched route."""
        from .admm import ADMM
        if self.batch != 1:
            raise NotImplementedError("projections given as Python callables run one problem at a time (batch=1); describe the "
                                      "set with projections.Box / ConvexSets for the batched device route")
        e = self.engine
        N, n, m = self.N, self.x_dim, self.u_dim
        Qr, Rr = self.compute_Rr_Qr(rho_x=rho_x if px is not None else None, rho_u=rho_u if pu is not None else None, dp=True)
        z_init = {}
        if batch_form:                                          # z <- unconstrained solution (sls.py:266-270)
            xs, us = self.solve_batch(x0)
            z_init = dict(z_x_init=np.asarray(xs).reshape(-1).copy(), z_u_init=np.asarray(us).reshape(-1).copy())
        self.solve_dp(Rr=Rr, Qr=Qr, xr=np.zeros(N * n), ur=np.zeros(N * m))
        last = None
        if batch_form and pu is not None:                       # the dense form's last control (see ADMM_LQT_Batch)
            R_, Rr_ = np.eye(m) * self.u_std, np.asarray(Rr[-1], dtype=np.float64)
            last = np.linalg.inv(R_ + Rr_) @ Rr_
        x0t = e._t(self._batched(x0, 1))
        one = torch.ones(1, dtype=e.dtype, device=e.device)
        stream = torch.cuda.current_stream().cuda_stream

        def f_argmin(x_reg, u_reg):
            if px is not None:
                e.zx.copy_(e._t(np.asarray(x_reg).reshape(1, N, n)))
            if pu is not None:
                e.zu.copy_(e._t(np.asarray(u_reg).reshape(1, N, m)))
            e.kern.riccati_ff(e.A, e.Bm, e.c0x, e.c0u, e.K, e.Quu, e.fac, e.Qux, e.k, Qr=e.Qr, Rr=e.Rr, zx=e.zx, lx=e.lx,
                              zu=e.zu, lu=e.lu, solve_mode=capi.SOLVE_INV, stream=stream)
            e.kern.rollout_ls(e.model, e.model_par, e.K, e.k, e.xhat, e.uhat, one, e.Qtab, e.ztab, e.seq, e.u_std, e.xx, e.xu,
                              x0=x0t, flags=capi.RO_ABSOLUTE, q_nonzero=e.q_nonzero, stream=stream)
            x, u = e.xx.cpu().numpy().astype(np.float64).reshape(-1), e.xu.cpu().numpy().astype(np.float64).reshape(-1)
            if last is not None:
                u[-m:] = last @ np.asarray(u_reg).reshape(N, m)[-1]
            return (x, u) if batch_form else (x, u, self._out(e.K), self._out(e.k))
        return ADMM(n * N, m * N, f_argmin, project_x=px if px is not None else False, project_u=pu if pu is not None else False,
                    alpha=alpha, max_iter=max_iter, tol=tol, log=log, **z_init)

    def _admm_lqt(self, x0, project_x, project_u, max_iter, rho_x, rho_u, alpha, tol, log, batch_form):
        e = self.engine
        B, N, n, m = self.batch, self.N, self.x_dim, self.u_dim
        px, pu = self._projection(project_x, n), self._projection(project_u, m)
        if any(p_ is not None and not isinstance(p_, (Box, ConvexSets)) for p_ in (px, pu)):
            return self._admm_lqt_host(x0, px, pu, max_iter, rho_x, rho_u, alpha, tol, log, batch_form)
        Qr, Rr = self.compute_Rr_Qr(rho_x=rho_x if px is not None else None, rho_u=rho_u if pu is not None else None, dp=True)
        self.solve_dp(Rr=Rr, Qr=Qr, xr=np.zeros(N * n), ur=np.zeros(N * m))
        sets = {}
        for blk, p_, d in (("x", px, n), ("u", pu, m)):
            if isinstance(p_, Box):
                lo, hi = (e._t(v) for v in p_.bounds(N, d))
                setattr(e, blk + "_lo", lo), setattr(e, blk + "_hi", hi)
            elif isinstance(p_, ConvexSets):                    # project_set_convex over the time steps (ISLS_PROJ_SETS)
                work = torch.zeros(B, N, d, dtype=e.dtype, device=e.device)
                wrap = lambda a: (torch.as_tensor(a, device=e.device) if a.dtype.kind in "iu" else e._t(a))   # noqa: E731
                desc = capi.Kernels.project_args_chain(work, work, p_.stages(), wrap=wrap)
                sets.update({blk + "_sets": desc, blk + "_col0": p_.cols[0], blk + "_work": work})
        x0t = e._t(self._batched(x0, 1))
        one = torch.ones(1, dtype=e.dtype, device=e.device)
        stream = torch.cuda.current_stream().cuda_stream
        last = None
        if batch_form:
            # z <- unconstrained solution (sls.py:266-270); the caller's regularised factors are rebuilt right after
            xs, us = self.solve_batch(x0)
            self.solve_dp(Rr=Rr, Qr=Qr, xr=np.zeros(N * n), ur=np.zeros(N * m))
            if e.zx is not None:
                e.zx.copy_(e._t(np.asarray(xs).reshape(B, N, n)))
            if e.zu is not None:
                e.zu.copy_(e._t(np.asarray(us).reshape(B, N, m)))
                R_, Rr_ = np.eye(m) * self.u_std, np.asarray(Rr[-1], dtype=np.float64)
                last = e._t(np.ascontiguousarray((np.linalg.inv(R_ + Rr_) @ Rr_).T))      # u_{N-1} = (z - lmb)_{N-1} @ last
        e.res_prev.fill_(1e6)
        e.admm_active.fill_(1)
        e.admm_iters.zero_()
        logs, chunk = [], 16
        for j0 in range(0, max_iter, chunk):                    # host looks at the stop flags every `chunk` iterations
            nrun = min(chunk, max_iter - j0)
            buf = torch.zeros(nrun, B, 2, dtype=e.dtype, device=e.device)
            iters_before = e.admm_iters.clone()
            for j in range(nrun):
                e.kern.riccati_ff(e.A, e.Bm, e.c0x, e.c0u, e.K, e.Quu, e.fac, e.Qux, e.k, Qr=e.Qr, Rr=e.Rr, zx=e.zx,
                                  lx=e.lx, zu=e.zu, lu=e.lu, solve_mode=capi.SOLVE_INV, active=e.admm_active, stream=stream)
                e.kern.rollout_ls(e.model, e.model_par, e.K, e.k, e.xhat, e.uhat, one, e.Qtab, e.ztab, e.seq, e.u_std,
                                  e.xx, e.xu, x0=x0t, flags=capi.RO_ABSOLUTE, q_nonzero=e.q_nonzero,
                                  active=e.admm_active, stream=stream)
                if last is not None:                              # the dense form's last control (see ADMM_LQT_Batch)
                    on = e.admm_active.to(torch.bool).view(B, 1)
                    e.xu[:, N - 1] = torch.where(on, (e.zu[:, N - 1] - e.lu[:, N - 1]) @ last, e.xu[:, N - 1])
                e.kern.admm_update(e.xx, e.xu, e.res, zx=e.zx, lx=e.lx, zu=e.zu, lu=e.lu,
                                   x_lo=e.x_lo if isinstance(px, Box) else None, x_hi=e.x_hi if isinstance(px, Box) else None,
                                   u_lo=e.u_lo if isinstance(pu, Box) else None, u_hi=e.u_hi if isinstance(pu, Box) else None,
                                   relax=alpha, tol_abs=tol, tol_rel=tol, res_prev=e.res_prev, active=e.admm_active,
                                   iters=e.admm_iters, stream=stream, **sets)
                buf[j].copy_(e.res)
            done = (e.admm_iters - iters_before).cpu().numpy()
            bh = buf.cpu().numpy()
            for j in range(int(done.max())):
                logs.append(bh[j, 0] if B == 1 else bh[j])
            if not bool(e.admm_active.any().item()):
                break
        out = (self._out(e.xx).reshape(-1) if B == 1 else e.xx.cpu().numpy().reshape(B, -1),
               self._out(e.xu).reshape(-1) if B == 1 else e.xu.cpu().numpy().reshape(B, -1), self._out(e.K), self._out(e.k))
        return out + ((logs,) if log else ())

    # ---- system level synthesis (config 5) ---------------------------------------------------------------------------
    def _transfer(self):
        """Sw, Su of the LTI model, built on first use (sls_dense.transfer_matrices)."""
        if self._Sw is None:
            assert self.A is not None, "Set the linear dynamics model by self.AB = [A,B] before calling this method."
            self._Sw, self._Su = dense.transfer_matrices(self.A, self.B, self.N)
        return self._Sw, self._Su

    Sw = property(lambda self: self._transfer()[0])
    Su = property(lambda self: self._transfer()[1])

    def _dense_cost(self):
        return dense.dense_cost(self.zs, self.Qs, self.seq, self.u_std, self.N, self.x_dim, self.u_dim)

    def compute_inverses(self, M):
        return dense.compute_inverses(np.asarray(M, dtype=np.float64), self.u_dim, self.N)

    def solve_sls(self, verbose=False):
        """Unconstrained system level synthesis (isls/sls.py:205-233): feed-forward du [N m] (one per problem of the batch)
        and the block-lower-triangular feedback map PHI_U [N m, N n]."""
        assert self.Q is not None, "Set the quadratic cost model by self.set_cost_variables() before calling this method."
        Sw, Su = self._transfer()
        Q, R, xd = self._dense_cost()
        PHI_U, du, self.l_side_invs = dense.solve_sls(Sw, Su, Q, R, xd, self.N, self.x_dim, self.u_dim, self.l_side_invs)
        return PHI_U, du

    def controller(self, PHI_U, du):
        """K = Phi_u Phi_x^-1, k = (I - K Su) du (isls/sls.py:235-242); PHI_U / du may carry a leading batch axis."""
        Sw, Su = self._transfer()
        PHI_U, du = np.asarray(PHI_U, dtype=np.float64), np.asarray(du, dtype=np.float64)
        if PHI_U.ndim == 3:
            Ks, ks = zip(*(dense.controller(Sw, Su, P_, d_) for P_, d_ in zip(PHI_U, du)))
            return np.stack(Ks), np.stack(ks)
        return dense.controller(Sw, Su, PHI_U, du)

    def ADMM_SLS(self, project_x=False, project_u=False, max_iter=5000, rho_x=0., rho_u=0., alpha=1., tol=1e-3,
                 verbose=False, log=False, rel_tol=1e-2):
        """SLS-ADMM with robust (chance) constraints on the controls w.r.t. the initial position (isls/sls.py:319-454).
        `project_u` is a `projections.ConvexSets` acting on the rows y = [d_u, phi_u] of the (N m) x (1 + n/2) variable
        (e.g. `projections.chance_constraint_rows`); its A / b / par arrays may carry a leading batch axis for problems
        that differ in bound or variance, and `zs` may be given per problem.  The set-up (transfer matrices, the (N m)^2
        inverse) is host numpy, the ADMM loop of all problems is one device launch.  With `project_x` (a `ConvexSets` over
        the rows [d_x, phi_x], or the reference's callable on the (N n) x (1 + n/2) array), a non-zero `rho_x`, or callables,
        the iteration runs over the feedback columns with the Riccati kernels instead.  Returns du, phi_u[, logs].
        `rel_tol` (not in the reference's signature) is the threshold of its second stop rule, hard-coded to 1e-2 there
        (sls.py:426); 0 disables it, which pins the iteration count (that rule fires on rounding noise)."""
        has_rho_x = rho_x is not None and np.any(np.asarray(rho_x) != 0)
        if project_x or has_rho_x or not isinstance(project_u, ConvexSets):
            # state constraints, state weights or projections given as the reference's callables: ADMM over the feedback
            # columns with the Riccati kernels (robust.py) instead of the one-launch kernel with the dense inverse
            from .robust import admm_sls_columns
            self.l_side_invs = None
            PHI_U, _ = self.solve_sls()
            du, phi_u, lg, iters = admm_sls_columns(self, project_x, project_u, max_iter, rho_x, rho_u, alpha, tol, PHI_U)
            self.sls_iters = iters
            if self.batch == 1:
                du, phi_u = du[0], phi_u[0]
            if not log:
                return du, phi_u
            return du, phi_u, ([lg[j, 0] for j in range(int(iters[0]))] if self.batch == 1 else lg)
        e = self.engine
        B, N, n, m = self.batch, self.N, self.x_dim, self.u_dim
        p = n // 2
        if project_u.dim != p + 1 or project_u.cols != (0, p + 1):
            raise ValueError(f"project_u must act on rows of dimension 1 + x_dim/2 = {p + 1}")
        self.l_side_invs = None
        PHI_U, _ = self.solve_sls()
        Sw, Su = self._transfer()
        Q, R, xd = self._dense_cost()
        rr = dense.rho_diagonal(rho_u, N, m)
        l_side_inv, r_side = dense.admm_sls_setup(Sw, Su, Q, R, xd, rr, p, B)
        dev = lambda a: e._t(np.ascontiguousarray(a))                                         # noqa: E731
        sets = [{k: (dev(v) if isinstance(v, np.ndarray) else v) for k, v in st.items()} for st in project_u.sets]
        x_u = torch.zeros(B, N * m, p + 1, dtype=e.dtype, device=e.device)
        logs = torch.full((B, int(max_iter), 2), float("nan"), dtype=e.dtype, device=e.device)
        iters = torch.zeros(B, dtype=torch.int32, device=e.device)
        e.kern.sls_admm(dev(l_side_inv), dev(r_side), dev(rr), sets, x_u, alpha=alpha, tol=tol, max_iter=max_iter,
                        rho=project_u.rho, inner_max_iter=project_u.max_iter, threshold=project_u.threshold, logs=logs,
                        iters=iters, rel_tol=rel_tol, stream=torch.cuda.current_stream().cuda_stream)
        xu = x_u.cpu().numpy().astype(np.float64)
        self.sls_iters = iters.cpu().numpy()
        du = xu[..., 0]
        phi_u = np.concatenate([xu[..., 1:], np.broadcast_to(PHI_U[:, p:], (B,) + PHI_U[:, p:].shape)], axis=-1)
        if B == 1:
            du, phi_u = du[0], phi_u[0]
        if not log:
            return du, phi_u
        lg = logs.cpu().numpy()
        return du, phi_u, ([lg[0, j] for j in range(int(self.sls_iters[0]))] if B == 1 else lg)

    def get_trajectory_sls(self, x0, K, k, noise_scale=0):
        """Closed loop with the dense causal SLS controller: u_i = (K x_{0..i} + k)_i, x_{i+1} = A x_i + B u_i for a set
        of initial states x0 [M, n] (isls/sls_base.py:91-105) -- the Monte-Carlo evaluation of the notebooks, one device
        thread per initial state."""
        if noise_scale:                                            # isls/sls_base.py:91-105 on the host (see get_trajectory_dp)
            from . import hostpath
            K, k = np.asarray(K, dtype=np.float64), np.asarray(k, dtype=np.float64)
            A, Bm = np.asarray(self.A, dtype=np.float64), np.asarray(self.B, dtype=np.float64)
            n_, m_ = self.x_dim, self.u_dim

            def control(i, x_log):
                xv = np.zeros((x_log.shape[0], self.N * n_))
                xv[:, :(i + 1) * n_] = x_log[:, :i + 1].reshape(x_log.shape[0], -1)
                return (xv @ K.T + k)[:, i * m_:(i + 1) * m_]
            return hostpath.noisy_closed_loop(lambda x, u: x @ A.T + u @ Bm.T, np.atleast_2d(np.asarray(x0, dtype=np.float64)),
                                              self.N, m_, control, noise_scale)
        e = self.engine
        x0 = np.atleast_2d(np.asarray(x0, dtype=np.float64))
        M, N, n, m = x0.shape[0], self.N, self.x_dim, self.u_dim
        dev = lambda a: e._t(np.ascontiguousarray(a))                                         # noqa: E731
        x_log = torch.zeros(M, N, n, dtype=e.dtype, device=e.device)
        u_log = torch.zeros(M, N, m, dtype=e.dtype, device=e.device)
        e.kern.sls_closed_loop(dev(self.A), dev(self.B), dev(np.asarray(K)), dev(np.asarray(k)), dev(x0), x_log, u_log,
                               stream=torch.cuda.current_stream().cuda_stream)
        return x_log.cpu().numpy().astype(np.float64), u_log.cpu().numpy().astype(np.float64)

    def get_trajectory_batch(self, x0, us, noise_scale=0):
        """Open loop: the control sequence us [N,m] applied from every initial state x0 [M,n] (isls/sls_base.py:61-74)."""
        K = np.zeros((self.N, self.u_dim, self.x_dim))
        return self.get_trajectory_dp(x0, K, np.asarray(us, dtype=np.float64).reshape(self.N, self.u_dim), noise_scale)

    # ---- at optimality / replanning: dense host algebra on the transfer matrices, as in the reference (cold paths) ----------
    def u_optimal(self, x0, PHI_U, du):
        """isls/sls_base.py:55-56."""
        return (np.asarray(PHI_U)[:, :self.x_dim] @ x0 + du).reshape(self.N, -1)[:-1]

    def x_optimal(self, x0, PHI_X, dx):
        """isls/sls_base.py:58-59."""
        return (np.asarray(PHI_X)[:, :self.x_dim] @ x0 + dx).reshape(self.N, -1)

    def initialize_replanning_procedure(self, K):
        """Map from a change of the stacked targets to the change of the feed-forward term (isls/sls.py:244-245)."""
        Sw, Su = self._transfer()
        Q, R, _ = self._dense_cost()
        SuTQ = Su.T @ Q
        self.replan_matrix = (np.eye(Su.shape[1]) - np.asarray(K) @ Su) @ np.linalg.solve(SuTQ @ Su + R, SuTQ)

    def replan_feedforward(self, k, xd):
        """isls/sls.py:247-248."""
        return k + self.replan_matrix.dot(xd - self._dense_cost()[2])
