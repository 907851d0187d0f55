"""Batched iLQR / DP-form iLQR-ADMM front end with the reference's `iSLS` class surface.

Reference: `iSLS` (isls/isls.py) + `iSLSBase` (isls/isls_base.py).  The public names, arguments and loop
semantics follow the reference (HEAD names plus the notebook-era aliases listed in SURVEY 8b); every
numerical step runs in the HIP kernels behind `engine.Engine`:

    backward_pass_DP  -> riccati_gain + riccati_ff           (isls/isls.py:229-308)
    rollout_DP        -> rollout_ls                          (isls/isls.py:310-334)
    iterate_once_dp   -> gain, ff, rollout_ls(NaN rule, acceptance test)   (isls/isls.py:336-374)
    ilqr_admm         -> DP form: ilqr_admm_outer + accept_step            (isls/isls.py:379-501, "TODO: add dp solution")
    isls_admm         -> DP form: gain + C x ff + columns_rollout + rollout_ls + columns_admm (isls/isls.py:503-712)
    backward_pass_batch / iterate_once_batch -> gain + ff + columns_rollout (column 0) + open-loop rollout_ls (isls.py:156-228)

Differences from the reference, all deliberate and documented in DESIGN.md: `ilqr_admm` uses the DP (Riccati)
solve instead of the dense batch-form least squares (identical iterates except the never-applied last control,
SURVEY 8a quirk i); forward models and costs are the built-in device implementations (`isls.models`, via-point
quadratic cost); status is per trajectory; nothing falls back to the CPU.
"""
import numpy as np
import torch

from . import _capi as capi
from . import hostpath
from .admm import ADMM
from .base import ALPHAS, Base
from .models import Model
from .projections import Box, ConvexSets
from .robust import isls_admm as _isls_admm


class iSLS(Base):
    def __init__(self, x_dim, u_dim, N, batch=1, dtype=np.float64, device="cuda"):
        super().__init__(x_dim, u_dim, N, batch=batch, dtype=dtype, device=device)
        self.alphas = ALPHAS.copy()                             # 10**linspace(0,-5,50), isls/isls_base.py:10-11
        self._forward_model = None
        self._cost_function = None
        self.cost_log = []
        self._K = self._k = None
        self._user_AB = False
        self._host_model = self._host_cost = False              # plain Python callables: the line search runs on the host

    @property
    def _dx_columns(self):
        """[d_x, phi_x] rows [B, N n, 1 + dim] of the last `isls_admm` x-step, copied from the device on first use"""
        if getattr(self, "_dx_columns_host", None) is None and getattr(self, "_dx_columns_dev", None) is not None:
            from .robust import ColumnSolver
            self._dx_columns_host = ColumnSolver.rows_to_host(self._dx_columns_dev)
        return getattr(self, "_dx_columns_host", None)

    # ---- setters / getters (isls/isls_base.py:74-158) ------------------------------------------------------
    @property
    def forward_model(self):
        return self._forward_model

    @forward_model.setter
    def forward_model(self, model):
        """An `isls.models` descriptor selects the device implementation of the rollout.  A plain callable
        `f(x[R,n], u[R,m]) -> [R,n]` (the reference's convention, isls/isls.py:153,332) is accepted too: the Riccati passes
        and the ADMM update stay on the GPU, the line-search rollouts go through the callable on the host (hostpath.py)."""
        if isinstance(model, Model):
            if model.x_dim != self.x_dim or model.u_dim != self.u_dim:
                raise ValueError("model dimensions do not match x_dim/u_dim")
            self._forward_model, self._host_model = model, False
            self.engine.set_model(model.model_id, model.params())
            return
        if not callable(model):
            raise TypeError("forward_model must be an isls.models descriptor or a callable f(x, u)")
        self._forward_model, self._host_model = model, True

    @property
    def cost_function(self):
        return self._cost_function if self._cost_function is not None else self.compute_cost

    @cost_function.setter
    def cost_function(self, function):
        """None -> the via-point quadratic cost of set_cost_variables; an `isls.costs` object (PseudoHuber) -> that cost
        on the device, for the line search and for the expansion (the reference's cost_function / get_Cs pair)."""
        if function is None:
            self._cost_function, self._host_cost = None, False
            return
        self._cost_function = function
        if hasattr(function, "cost_model"):
            self._host_cost = False
            self.engine.set_cost_model(function.cost_model, function.params())
        elif callable(function):
            # the reference's `cost_function(x[L,N,n], u[L,N,m]) -> [L]` (isls.py:360): evaluated on the host for every
            # candidate of the line search; its expansion comes from the caller's get_Cs
            self._host_cost = True
        else:
            raise TypeError("cost_function must be None, an isls.costs object or a callable cost(x, u)")

    @property
    def AB(self):
        return [self.A, self.B]

    @AB.setter
    def AB(self, value):
        """A [N,n,n] / [n,n] (or with a leading batch axis), B likewise: linearisation used by the next backward pass."""
        A, B = np.asarray(value[0], dtype=np.float64), np.asarray(value[1], dtype=np.float64)
        self.A, self.B = A, B
        e = self.engine
        # a linearisation of the same shape as the one in use is copied INTO the buffers the engine already holds: argument blocks
        # and captured HIP graphs (isls_admm) keep raw device pointers, and a get_AB callback delivers a fresh array per outer
        # iteration -- replacing the tensors would leave those pointers on freed memory
        for name, new in (("A", e._t(A)), ("Bm", e._t(B))):
            old = getattr(e, name)
            if old is not None and old.shape == new.shape and old.is_contiguous():
                old.copy_(new)
            else:
                setattr(e, name, new)
                e._outer_args = None
        e.ab_from_caller()
        self._user_AB = True

    @property
    def nominal_values(self):
        return self.x_nom, self.u_nom

    @nominal_values.setter
    def nominal_values(self, value):
        e = self.engine
        if e.Qtab is None:                                      # no via-point cost (a callable cost only): a zero table
            e.set_quadratic_cost(np.zeros((1, self.x_dim)), np.zeros((1, self.x_dim, self.x_dim)), np.zeros(self.N, dtype=np.int32), 0.0)
        e.set_nominal(self._batched(value[0], 2), self._batched(value[1], 2))
        if self._host_cost:                                     # the nominal's cost through the caller's function
            self._refresh_host_cost(reset_history=True)
        self.cost_log.append(self.cost)

    def _refresh_host_cost(self, reset_history=False):
        e = self.engine
        e.cost.copy_(e._t(hostpath.nominal_cost(self)))
        if reset_history:
            e.cost_hist[:, 0] = e.cost

    @property
    def _host_ls(self):
        return self._host_model or self._host_cost

    def _line_search(self, L, flags=0, active=None):
        """Line search over alphas[:L]: the rollout kernel, or the host route when the model / cost is a Python callable."""
        if self._host_ls:
            hostpath.line_search(self, L, flags, active)
        else:
            self.engine.rollout(L, flags=flags, active=active)

    @property
    def x_nom(self):
        return self._out(self.engine.xhat)

    @property
    def u_nom(self):
        return self._out(self.engine.uhat)

    @property
    def cost(self):
        c = self.engine.cost.detach().cpu().numpy()
        return float(c[0]) if self.batch == 1 else c

    @property
    def K(self):
        return self._out(self.engine.K)

    @property
    def k(self):
        return self._out(self.engine.k)

    @property
    def status(self):
        """Per-trajectory status bits (ISLS_ST_*): non-PD Quu, NaN cost, line search rejected."""
        return self.engine.status.cpu().numpy()

    def reset(self):
        self.cost_log = []
        self.engine.status.zero_()
        self.engine.outer_active.fill_(1)

    def compute_cost(self, x, u=None):
        """(x-xd)'Q(x-xd) + u'Ru, no 1/2 (SLSBase.compute_cost, isls/sls_base.py:25-44) for arrays [.., N, n]."""
        x = np.asarray(x, dtype=np.float64)
        lead = x.shape[:-2]
        seq = self.seq
        zs = self.zs if self.zs.ndim == 2 else None
        if zs is None:
            raise NotImplementedError("compute_cost on host needs shared via-points; use .cost for per-trajectory targets")
        dx = x - zs[seq]
        c = np.einsum("...ti,tij,...tj->...", dx, self.Qs[seq], dx)
        if u is not None:
            u = np.asarray(u, dtype=np.float64)
            c = c + self.u_std * np.sum(u * u, axis=(-1, -2))
        return float(c) if lead == () else c

    # ---- DP iLQR kernels -----------------------------------------------------------------------------------
    def _linearize(self, get_AB):
        e = self.engine
        model = self._forward_model
        if get_AB is None or (getattr(get_AB, "__self__", None) is model and not self._host_model):
            if model is None or self._host_model:
                raise ValueError("set forward_model to an isls.models descriptor or pass get_AB (a callable forward model has no "
                                 "built-in linearisation)")
            if self._user_AB:                                   # restore the engine's own dense buffers
                z = lambda *s: torch.zeros(*s, dtype=e.dtype, device=e.device)   # noqa: E731
                e.A, e.Bm = z(e.B, e.N, e.n, e.n), z(e.B, e.N, e.n, e.m)
                self._user_AB = False
            e.linearize()
            return
        # user callback in the reference's convention (numpy, one trajectory): host round trip by design
        xs, us = e.xhat.cpu().numpy(), e.uhat.cpu().numpy()
        AB = [get_AB(xs[b], us[b]) for b in range(self.batch)]
        self.AB = np.stack([np.array(a[0]) for a in AB]), np.stack([np.array(a[1]) for a in AB])

    def _expand(self, Cts=None, cts=None):
        e = self.engine
        if Cts is None:
            e.Cux = None
            e.expand()
            return
        n = self.x_dim
        e.allow_shared_hessian = False                          # the caller's Hessians differ per trajectory
        Cts, cts = self._batched(Cts, 3), self._batched(cts, 2)
        e.Cxx.copy_(e._t(Cts[..., :n, :n])), e.Cuu.copy_(e._t(Cts[..., n:, n:]))
        e.Cux = e._t(Cts[..., n:, :n])
        e.c0x.copy_(e._t(cts[..., :n])), e.c0u.copy_(e._t(cts[..., n:]))

    def backward_pass_DP(self, Cts=None, cts=None):
        """Riccati recursion about the nominal (isls/isls.py:229-308).  Returns (K [N,m,n], k [N,m])."""
        if not self._user_AB:
            self._linearize(None)                               # built-in model: A_t, B_t along the current nominal
        self._expand(Cts, cts)
        self.engine.gain()
        self.engine.feedforward()
        return self.K, self.k

    def rollout_DP(self, K, k):
        """Closed-loop rollout of the candidates k[l] (isls/isls.py:310-334): returns (x_log [L,N,n], u_log [L,N,m])."""
        e = self.engine
        k = np.asarray(k, dtype=np.float64)                     # [L,N,m] (batch == 1) or [B,L,N,m]
        if self._host_model:
            if self.batch != 1:
                raise NotImplementedError("rollout_DP through a callable forward model: batch == 1")
            K = np.asarray(K, dtype=np.float64)
            xn, un, f = self.x_nom, self.u_nom, self._forward_model
            x = np.tile(xn[0], (k.shape[0], 1))
            x_log, u_log = np.zeros((k.shape[0], self.N, self.x_dim)), np.zeros((k.shape[0], self.N, self.u_dim))
            for i in range(self.N):
                u = (x - xn[i]) @ K[i].T + k[:, i] + un[i]
                u_log[:, i], x_log[:, i] = u, x
                x = np.asarray(f(x, u), dtype=np.float64)
            return x_log, u_log
        if k.ndim == 3 and self.batch > 1:
            k = np.broadcast_to(k[None], (self.batch,) + k.shape)
        Kt = e._t(self._batched(K, 3))
        xs, us = [], []
        one = torch.ones(1, dtype=e.dtype, device=e.device)
        for l in range(k.shape[-3]):
            kl = e._t(self._batched(k[l] if k.ndim == 3 else k[:, l], 2))
            e.kern.rollout_ls(e.model, e.model_par, Kt, kl, e.xhat, e.uhat, one, e.Qtab, e.ztab, e.seq, e.u_std, e.xx, e.xu,
                              q_nonzero=e.q_nonzero, stream=torch.cuda.current_stream().cuda_stream)
            xs.append(e.xx.cpu().numpy()), us.append(e.xu.cpu().numpy())
        x, u = np.stack(xs, axis=1), np.stack(us, axis=1)
        return (x[0], u[0]) if self.batch == 1 else (x, u)

    def iterate_once_dp(self, max_line_search=15, verbose=False, Cts=None, cts=None, _linearized=False):
        """Backward pass + line search over alphas[:max_line_search] with the NaN rule and the acceptance test
        (isls/isls.py:336-374).  Returns (fp_success, K, k); fp_success is a bool, or a bool array when batched."""
        e = self.engine
        if not _linearized and not self._user_AB:
            self._linearize(None)
        self._expand(Cts, cts)
        e.status.zero_()
        e.gain(active=e.outer_active)
        e.feedforward(active=e.outer_active)
        self._line_search(max_line_search, flags=capi.RO_NAN_TO_1E5 | capi.RO_ACCEPT_TEST, active=e.outer_active)
        st = e.status.cpu().numpy()
        ok = (st & capi.ST_LS_REJECT) == 0
        if (st & capi.ST_NOT_PD).any():
            raise np.linalg.LinAlgError(f"Quu not positive definite for trajectories {np.nonzero(st & capi.ST_NOT_PD)[0].tolist()}")
        e.accept_x_step()                                       # rejected trajectories received their nominal back
        if self.batch == 1:
            if ok[0]:
                self.cost_log.append(self.cost)
            elif verbose:
                print("Forward pass failed with a cost of", float(e.cost_new[0]))
            return bool(ok[0]), self.K, self.k
        self.cost_log.append(self.cost)
        return ok, self.K, self.k

    def solve(self, get_AB=None, get_Cs=None, is_dynamics_linear=False, is_cost_quadratic=False, method='dp',
              max_iter=100, max_line_search_iter=25, tol_fun=1e-5, tol_grad=1e-4, verbose=False):
        """iLQR outer loop with the reference's stop rules (isls/isls.py:54-132), per trajectory when batched."""
        if method not in ('dp', 'batch'):
            raise NotImplementedError("method must be 'dp' or 'batch' (the reference raises for 'sls' too, isls.py:121-122)")
        e = self.engine
        e.outer_active.fill_(1)
        prev = np.atleast_1d(np.array(self.cost, dtype=np.float64)).copy()
        Cts = cts = None
        for i in range(max_iter):
            if verbose:
                print("Iteration", i)
            if not (is_dynamics_linear and i > 0):
                self._linearize(get_AB)
            Cts, cts = self._user_expansion(get_Cs) if not (is_cost_quadratic and i > 0) else (Cts, cts)
            if method == 'dp':
                ok, _, _ = self.iterate_once_dp(max_line_search=max_line_search_iter, verbose=verbose, Cts=Cts, cts=cts,
                                                _linearized=True)
            else:
                self._user_AB, keep = True, self._user_AB         # linearised above: backward_pass_batch must not redo it
                ok = self.iterate_once_batch(max_line_search=max_line_search_iter, Cts=Cts, cts=cts, verbose=verbose)
                self._user_AB = keep
            cur = np.atleast_1d(np.array(self.cost, dtype=np.float64))
            okv = np.atleast_1d(ok)
            act = e.outer_active.cpu().numpy().astype(bool)
            # reference: |diff(cost_log[-2:])| < tol_fun (a rejected step leaves cost_log untouched, so the test
            # is on the last two ACCEPTED costs), then `not fp_success`
            small = np.abs(cur - prev) < tol_fun
            stop = act & ((okv & small) | ~okv)
            prev = np.where(act & okv, cur, prev)
            act &= ~stop
            e.outer_active.copy_(torch.as_tensor(act.astype(np.int32), device=e.device))
            if not act.any():
                if verbose:
                    print("Converged / stopped at iteration", i + 1)
                break
        return None

    def _device_expansion(self, get_Cs):
        """True when the device expands the cost itself: no get_Cs, or the get_Cs of the isls.costs object on the device."""
        return get_Cs is None or (self._cost_function is not None and not self._host_cost and
                                  getattr(get_Cs, "__self__", None) is self._cost_function)

    def _user_expansion(self, get_Cs):
        """(Cts, cts) from the caller's derivative callback `cts, Cts = get_Cs(x_nom, u_nom)` (isls/isls.py:102), evaluated
        on the host per trajectory -- or (None, None) when the device expands the cost itself."""
        if self._device_expansion(get_Cs):
            if self._host_cost and get_Cs is None:
                raise ValueError("a callable cost_function needs get_Cs (its gradient / Hessian along the nominal)")
            return None, None
        return hostpath.expansion(self, get_Cs)

    def _check_get_Cs(self, get_Cs):
        if self._host_cost and get_Cs is None:
            raise ValueError("a callable cost_function needs get_Cs (its gradient / Hessian along the nominal)")

    def solve_ilqr(self, get_AB=None, max_ilqr_iter=100, max_line_search_iter=25, dp=True, verbose=False, **kw):
        """Notebook-era name (Car notebooks :254): iLQR with the quadratic cost set by set_cost_variables."""
        return self.solve(get_AB, method='dp' if dp else 'batch', max_iter=max_ilqr_iter, max_line_search_iter=max_line_search_iter,
                          verbose=verbose, **kw)

    # ---- iLQR-ADMM (DP form) --------------------------------------------------------------------------------
    def ilqr_admm(self, get_AB=None, get_Cs=None, project_x=False, project_u=False, max_iter=20,
                  max_line_search_iter=20, max_admm_iter=20, rho_x=None, rho_u=None, alpha=1, tol=1e-3,
                  verbose=False, log=False, k_max=None, max_line_search=None, threshold=None):
        """Outer loop of isls/isls.py:420-499 with the Riccati (DP) inner solve.  Per outer iteration:
        linearise, expand, then max_admm_iter x [ff pass, line search over alphas[:max_line_search_iter] without
        acceptance test, z/lambda update] with lambda reset and z warm-started, nominal <- last x-step, and the
        stop rules |dcost| < 1e-3 / oscillation < 1e-3.  Returns the residual log of the last outer iteration."""
        if self._host_cost and get_Cs is None:
            raise ValueError("a callable cost_function needs get_Cs (its gradient / Hessian along the nominal)")
        max_iter = k_max if k_max is not None else max_iter
        L = max_line_search if max_line_search is not None else max_line_search_iter
        tol = threshold if threshold is not None else tol
        e = self.engine
        px, pu, host_proj = self._setup_admm(project_x, project_u, rho_x, rho_u, alpha)
        J = int(max_admm_iter)
        logbuf = torch.zeros(J, self.batch, 2, dtype=e.dtype, device=e.device)
        e.outer_active.fill_(1)
        logs = []
        for j in range(max_iter):
            self._linearize(get_AB)
            self._expand_regularised(get_Cs)
            if host_proj:
                logs = self._admm_host(px, pu, L, J, tol, alpha, verbose)
            else:
                e.build_outer(L, J, tol_abs=tol, tol_rel=tol, log=logbuf)
                e.run_outer()
                lb = logbuf.cpu().numpy()
                logs = [lb[i, 0] if self.batch == 1 else lb[i] for i in range(J)]
            e.accept_x_step(tol_cost=1e-3, tol_osc=1e-3)
            self.cost_log.append(self.cost)
            st = e.status.cpu().numpy()
            if (st & capi.ST_NOT_PD).any():
                raise np.linalg.LinAlgError("Quu not positive definite")
            if verbose:
                print("Iteration number ", j, "iLQR cost: ", self.cost)
            if not bool(e.outer_active.any().item()):
                break
        # the reference's `logs` holds one [prim, dual] entry per EXECUTED ADMM iteration of the last outer
        # iteration; batched: all J rows (rows after a trajectory's own stop repeat its last residuals) and
        # `self.admm_iters` tells how many are real per trajectory
        self.admm_iters = e.admm_iters.cpu().numpy()
        return logs[:int(self.admm_iters[0])] if self.batch == 1 and not host_proj else logs

    def _setup_admm(self, project_x, project_u, rho_x, rho_u, alpha):
        """project_x / project_u of an ilqr_admm call -> the engine's ADMM state (weights, boxes or device sets, z / lambda
        buffers); returns (px, pu, host_proj): the projections as objects and whether the z-step runs through the host."""
        e = self.engine
        px, pu = self._projection(project_x, self.x_dim), self._projection(project_u, self.u_dim)
        on_device = (Box, ConvexSets)
        host_proj = (px is not None and not isinstance(px, on_device)) or (pu is not None and not isinstance(pu, on_device))
        host_proj = host_proj or self._host_ls                  # a host line search: the whole ADMM loop is host driven
        xs = px if isinstance(px, ConvexSets) and not host_proj else None
        us = pu if isinstance(pu, ConvexSets) and not host_proj else None
        xb = px.bounds(self.N, self.x_dim) if isinstance(px, Box) else ((-np.inf, np.inf) if px is not None and xs is None else None)
        ub = pu.bounds(self.N, self.u_dim) if isinstance(pu, Box) else ((-np.inf, np.inf) if pu is not None and us is None else None)
        e.set_admm(rho_x=rho_x if px is not None else None, rho_u=rho_u if pu is not None else None,
                   x_box=xb, u_box=ub, relax=alpha, x_sets=xs, u_sets=us)
        return px, pu, host_proj

    def _expand_regularised(self, get_Cs):
        """Quadratic expansion about the nominal plus the ADMM regulariser's Hessians: on the device for the built-in costs,
        from the caller's get_Cs otherwise (gradients of the regulariser are added by the feed-forward pass either way)."""
        e = self.engine
        if self._device_expansion(get_Cs):
            e.Cux = None
            e.expand()
            return
        Cts, cts = hostpath.expansion(self, get_Cs)
        self._expand(Cts, cts)
        if e.Qr is not None:
            e.Cxx.add_(2.0 * (e.Qr if e.Qr.ndim == 4 else e.Qr.unsqueeze(0)))
        if e.Rr is not None:
            e.Cuu.add_(2.0 * (e.Rr if e.Rr.ndim == 4 else e.Rr.unsqueeze(0)))

    def _admm_host(self, px, pu, L, J, tol, alpha, verbose):
        """Generic route for projections without a device kernel: x-step on the GPU, z-step through the caller's
        numpy functions, one trajectory at a time (isls/admm.py semantics via `isls.admm.ADMM`)."""
        e = self.engine
        B, N, n, m = self.batch, self.N, self.x_dim, self.u_dim
        e.gain()
        res = []
        # the B trajectories advance in lock-step: every ADMM iteration launches the batched x-step once;
        # lambda restarts at zero and z is warm-started from the previous outer iteration (isls.py:414-417,489-490)
        z_x = e.zx.cpu().numpy().reshape(B, -1).copy() if e.zx is not None else None
        z_u = e.zu.cpu().numpy().reshape(B, -1).copy() if e.zu is not None else None
        l_x = np.zeros_like(z_x) if z_x is not None else None
        l_u = np.zeros_like(z_u) if z_u is not None else None
        prev = np.full((B, 2), 1e6)
        active = np.ones(B, dtype=bool)
        for it in range(J):
            if z_x is not None:
                e.zx.copy_(e._t(z_x.reshape(B, N, n))), e.lx.copy_(e._t(l_x.reshape(B, N, n)))
            if z_u is not None:
                e.zu.copy_(e._t(z_u.reshape(B, N, m))), e.lu.copy_(e._t(l_u.reshape(B, N, m)))
            act_t = torch.as_tensor(active.astype(np.int32), device=e.device)
            e.feedforward(active=act_t)
            self._line_search(L, active=act_t)
            xx, xu = e.xx.cpu().numpy().reshape(B, -1), e.xu.cpu().numpy().reshape(B, -1)
            cur = np.zeros((B, 2))
            for b in range(B):
                if not active[b]:
                    cur[b] = prev[b]
                    continue
                prim = dual = 0.0
                for z, l, x, proj in ((z_x, l_x, xx, px), (z_u, l_u, xu, pu)):
                    if z is None:
                        continue
                    fn = proj if not isinstance(proj, Box) else proj.__call__
                    z_new = np.asarray(fn(alpha * x[b] + (1 - alpha) * z[b] + l[b]), dtype=np.float64).reshape(-1)
                    r = x[b] - z_new
                    l[b] += r
                    prim += np.linalg.norm(r)
                    dual += np.linalg.norm(z_new - z[b])
                    z[b] = z_new
                cur[b] = prim, dual
                stop = (prim < tol and dual < tol) or (abs(prev[b, 0] - prim) / (prev[b, 0] + 1e-30) < tol and
                                                       abs(prev[b, 1] - dual) / (prev[b, 1] + 1e-30) < tol)
                prev[b] = prim, dual
                if stop:
                    active[b] = False
            res.append(cur[0].copy() if B == 1 else cur.copy())
            if not active.any():
                break
        if z_x is not None:
            e.zx.copy_(e._t(z_x.reshape(B, N, n))), e.lx.copy_(e._t(l_x.reshape(B, N, n)))
        if z_u is not None:
            e.zu.copy_(e._t(z_u.reshape(B, N, m))), e.lu.copy_(e._t(l_u.reshape(B, N, m)))
        return res

    isls_admm = _isls_admm                                      # isls/isls.py:503-712, DP form (robust.py)

    def controller(self, PHI_U, du):
        """K = Phi_u Phi_x^-1, k = (I - K Su) du with the transfer matrices of the last linearisation (notebook-era
        `iSLS.controller`, isls/sls.py:235-242 on `Base.AB`'s Sw / Su): dense host set-up per problem, as in the reference.
        PHI_U [N m, N n] and du [N m] (leading batch axis when batched)."""
        from . import sls_dense as dense
        A, Bm = self.engine.A.cpu().numpy().astype(np.float64), self.engine.Bm.cpu().numpy().astype(np.float64)
        PHI_U, du = np.asarray(PHI_U, dtype=np.float64), np.asarray(du, dtype=np.float64)
        if self.batch == 1 and PHI_U.ndim == 2:
            PHI_U, du = PHI_U[None], du[None]
        Ks, ks = zip(*(dense.controller(*dense.transfer_matrices_ltv(np.broadcast_to(A[b], (self.N,) + A.shape[-2:]),
                                                                   np.broadcast_to(Bm[b], (self.N,) + Bm.shape[-2:])), PHI_U[b], du[b])
                       for b in range(self.batch)))
        return (Ks[0], ks[0]) if self.batch == 1 else (np.stack(Ks), np.stack(ks))

    def get_trajectory_sls(self, x0, K, k, noise_scale=0, problem=0):
        """Monte-Carlo closed loop of the dense controller about the nominal of problem `problem` through the forward model
        (isls/isls_base.py:28-42): x0 [M, n] -> (x_log [M,N,n], u_log [M,N,m]); one device thread per initial state."""
        e = self.engine
        if noise_scale or self._host_model:
            # process noise comes from numpy's global generator, one draw per step in the reference's order: host loop
            # through the (numpy-callable) forward model, isls/isls_base.py:28-42
            K, k = np.asarray(K, dtype=np.float64), np.asarray(k, dtype=np.float64)
            xn, un = e.xhat[problem].cpu().numpy().astype(np.float64), e.uhat[problem].cpu().numpy().astype(np.float64)
            n, m = self.x_dim, self.u_dim

            def control(i, x_log):
                xv = np.zeros((x_log.shape[0], self.N * n))
                xv[:, :(i + 1) * n] = (x_log[:, :i + 1] - xn[None, :i + 1]).reshape(x_log.shape[0], -1)
                return (xv @ K.T + k)[:, i * m:(i + 1) * m] + un[i]
            return hostpath.noisy_closed_loop(self._forward_model, x0, self.N, m, control, noise_scale)
        x0 = np.asarray(x0, dtype=np.float64)
        single = x0.ndim == 1
        x0 = np.atleast_2d(x0)
        M = x0.shape[0]
        dev = lambda a: e._t(np.ascontiguousarray(a))                           # noqa: E731
        par = e.model_par if e.model_par.ndim == 1 else e.model_par[problem].contiguous()
        x_log = torch.zeros(M, self.N, self.x_dim, dtype=e.dtype, device=e.device)
        u_log = torch.zeros(M, self.N, self.u_dim, dtype=e.dtype, device=e.device)
        e.kern.dense_closed_loop(e.model, par, dev(np.asarray(K)), dev(np.asarray(k)), dev(x0), x_log, u_log,
                                 xhat=e.xhat[problem].contiguous(), uhat=e.uhat[problem].contiguous(),
                                 stream=torch.cuda.current_stream().cuda_stream)
        x, u = x_log.cpu().numpy().astype(np.float64), u_log.cpu().numpy().astype(np.float64)
        return (x[0], u[0]) if single else (x, u)

    # ---- batch-form iLQR (isls/isls.py:135-228) through the Riccati kernels ---------------------------------------------
    def rollout_batch(self, x_nom, u_nom):
        """Open-loop rollouts from x_nom[0] (isls/isls.py:135-154): u_nom [L,N,m] (or [B,L,N,m]) -> (x_log, u_nom)."""
        e = self.engine
        u = np.asarray(u_nom, dtype=np.float64)
        ub = u if (self.batch > 1 and u.ndim == 4) else np.broadcast_to(u[None], (self.batch,) + u.shape)
        x0 = e._t(self._batched(np.asarray(x_nom, dtype=np.float64), 2)[:, 0].copy())
        zero_K = torch.zeros(self.batch, self.N, self.u_dim, self.x_dim, dtype=e.dtype, device=e.device)
        zero_u = torch.zeros(self.batch, self.N, self.u_dim, dtype=e.dtype, device=e.device)
        one = torch.ones(1, dtype=e.dtype, device=e.device)
        xs = []
        for l in range(ub.shape[1]):
            e.kern.rollout_ls(e.model, e.model_par, zero_K, e._t(np.ascontiguousarray(ub[:, l])), e.xhat, zero_u, one, e.Qtab, e.ztab,
                              e.seq, e.u_std, e.xx, e.xu, x0=x0, q_nonzero=e.q_nonzero, cost_model=e.cost_model, cost_par=e.cost_par,
                              stream=torch.cuda.current_stream().cuda_stream)
            xs.append(e.xx.cpu().numpy())
        x = np.stack(xs, axis=1)
        return (x[0] if self.batch == 1 else x), u

    def backward_pass_batch(self, Cts=None, cts=None):
        """delta_u_opt [N,m] of the batch-form least squares (isls/isls.py:156-189): the minimiser of the LQ sub-problem
        about the nominal, i.e. the Riccati solution rolled through the linearised dynamics, plus the dense form's last
        control (column 0 of isls_columns_rollout; SURVEY 8a quirk i)."""
        e = self.engine
        if not self._user_AB:
            self._linearize(None)
        self._expand(Cts, cts)
        e.gain()
        z = lambda *s_: torch.zeros(*s_, dtype=e.dtype, device=e.device)        # noqa: E731
        B, N, n, m = self.batch, self.N, self.x_dim, self.u_dim
        kcol, dx, du = z(2, B, N, m), z(2, B, N, n), z(2, B, N, m)
        e.kern.riccati_ff(e.A, e.Bm, e.c0x, e.c0u, e.K, e.Quu, e.fac, e.Qux, kcol[0], solve_mode=e.solve_mode,
                          stream=torch.cuda.current_stream().cuda_stream)
        e.kern.columns_rollout(e.A, e.Bm, e.hessians()[1], e.c0u, e.K, kcol, dx, du, stream=torch.cuda.current_stream().cuda_stream)
        self._du_batch = du[0]
        return self._out(du[0])

    def iterate_once_batch(self, verbose=False, max_line_search=15, **kwargs):
        """Batch-form backward pass + open-loop line search with the acceptance test costs[ind] < cost
        (isls/isls.py:191-225).  Returns fp_success (bool, or a bool array when batched)."""
        e = self.engine
        self.backward_pass_batch(**kwargs)
        zero_K = torch.zeros(self.batch, self.N, self.u_dim, self.x_dim, dtype=e.dtype, device=e.device)
        e.status.zero_()
        e.kern.rollout_ls(e.model, e.model_par, zero_K, self._du_batch, e.xhat, e.uhat, e.alphas[:max_line_search], e.Qtab, e.ztab,
                          e.seq, e.u_std, e.xx, e.xu, best=e.best, cost_new=e.cost_new, cost_cur=e.cost, flags=capi.RO_ACCEPT_TEST,
                          status=e.status, active=e.outer_active, q_nonzero=e.q_nonzero, cost_model=e.cost_model,
                          cost_par=e.cost_par, stream=torch.cuda.current_stream().cuda_stream)
        ok = (e.status.cpu().numpy() & capi.ST_LS_REJECT) == 0
        e.accept_x_step()
        if self.batch == 1:
            if ok[0]:
                self.cost_log.append(self.cost)
            return bool(ok[0])
        self.cost_log.append(self.cost)
        return ok

    # ---- closed-loop evaluation (isls/isls_base.py:28-71) -----------------------------------------------------
    def get_trajectory_batch(self, x0, us, noise_scale=0):
        """Open loop: the control sequence us [N,m] (or [B,N,m]) applied from x0 (isls/isls_base.py:44-57) -- the notebooks'
        way to turn initial controls into a nominal trajectory."""
        us = np.asarray(us, dtype=np.float64)
        return self.get_trajectory_dp(x0, np.zeros(us.shape[:-1] + (self.u_dim, self.x_dim)), us, noise_scale)

    def _transfer_ltv(self, problem=0):
        from . import sls_dense as dense
        A, Bm = self.engine.A.cpu().numpy().astype(np.float64), self.engine.Bm.cpu().numpy().astype(np.float64)
        b = min(problem, A.shape[0] - 1)
        return dense.transfer_matrices_ltv(np.broadcast_to(A[b], (self.N,) + A.shape[-2:]), np.broadcast_to(Bm[b], (self.N,) + Bm.shape[-2:]))

    # dense transfer matrices of the last linearisation (Base.AB's Sw / Su, isls/base.py:98-119; notebook-era names C / D):
    # host numpy, built on demand for problem 0 of the batch -- the solvers themselves never form them
    Sw = property(lambda self: self._transfer_ltv()[0])
    Su = property(lambda self: self._transfer_ltv()[1])
    C, D = Sw, Su

    def get_trajectory_dp(self, x0, K, k, noise_scale=0):
        """u_t = K_t x_t + k_t, x_{t+1} = f(x_t, u_t) from x0 (absolute form), noise-free only.  x0 [n] (or [B,n]): one
        trajectory per problem of the batch; x0 [M,n] with batch == 1: M initial states against the same controller."""
        e = self.engine
        if noise_scale or self._host_model:                          # isls/isls_base.py:59-71 on the host (see get_trajectory_sls)
            K, k = np.asarray(K, dtype=np.float64), np.asarray(k, dtype=np.float64)
            if K.ndim == 4 or k.ndim == 3:
                raise NotImplementedError("noisy / callable-model closed loops take one controller (K [N,m,n], k [N,m])")
            return hostpath.noisy_closed_loop(self._forward_model, x0, self.N, self.u_dim,
                                              lambda i, x_log: x_log[:, i] @ K[i].T + k[i], noise_scale)
        x0 = np.asarray(x0, dtype=np.float64)
        if x0.ndim == 2 and self.batch == 1 and x0.shape[0] != 1:      # Monte Carlo over initial states: a batch of M rollouts
            mc = iSLS(self.x_dim, self.u_dim, self.N, batch=x0.shape[0], dtype=self.np_dtype, device=e.device)
            mc.forward_model = self._forward_model
            mc.engine.Qtab, mc.engine.ztab, mc.engine.seq, mc.engine.q_nonzero = e.Qtab, e.ztab, e.seq, e.q_nonzero
            mc.engine.u_std, mc.engine.cost_model, mc.engine.cost_par = e.u_std, e.cost_model, e.cost_par
            return mc.get_trajectory_dp(x0, K, k)
        x0 = self._batched(x0, 1)
        if e.Qtab is None:                                       # no cost yet (the notebooks roll the initial controls out first)
            e.set_quadratic_cost(np.zeros((1, self.x_dim)), np.zeros((1, self.x_dim, self.x_dim)), np.zeros(self.N, dtype=np.int32), 0.0)
        one = torch.ones(1, dtype=e.dtype, device=e.device)
        e.kern.rollout_ls(e.model, e.model_par, e._t(self._batched(K, 3)), e._t(self._batched(k, 2)), e.xhat, e.uhat, one,
                          e.Qtab, e.ztab, e.seq, e.u_std, e.xx, e.xu, x0=e._t(x0), flags=capi.RO_ABSOLUTE,
                          q_nonzero=e.q_nonzero, stream=torch.cuda.current_stream().cuda_stream)
        return self._out(e.xx), self._out(e.xu)
