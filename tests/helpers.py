"""Shared test helpers: problem arrays for the configs and a numpy driver that sequences the ORACLE
kernels into the outer DP-form iLQR-ADMM loop (semantics of iSLS.ilqr_admm, isls/isls.py:420-499,
and of ADMM(), isls/admm.py:6-106).  Test infrastructure only."""
import numpy as np

import isls_problems as P
from isls import _capi as capi

ALPHAS = 10.0 ** np.linspace(0.0, -5.0, 50)     # isls/isls_base.py:10-11


def model_par(cfg, dtype=np.float64):
    if cfg["model"] == P.MODEL_LTI:
        return np.concatenate([cfg["A"].ravel(), cfg["B"].ravel()]).astype(dtype)
    return np.array([cfg["dt"]], dtype=dtype)


def problem_arrays(cfg, bsel, dtype=np.float64):
    """numpy inputs for the trajectories `bsel` of a batched config."""
    bsel = list(bsel)
    B, N, n, m = len(bsel), cfg["N"], cfg["n"], cfg["m"]
    d = dict(B=B, N=N, n=n, m=m, model=cfg["model"], u_std=cfg["u_std"])
    xh = np.zeros((B, N, n))
    uh = np.zeros((B, N, m))
    for i, b in enumerate(bsel):
        xh[i], uh[i] = P.initial_nominal(cfg, b)
    d["xhat"], d["uhat"] = xh.astype(dtype), uh.astype(dtype)
    zs = cfg["zs"]
    d["ztab"] = (zs[bsel] if zs.ndim == 3 else zs).astype(dtype).copy()
    d["Qtab"] = cfg["Qs"].astype(dtype).copy()
    d["seq"] = cfg["seq"].astype(np.int32).copy()
    d["model_par"] = model_par(cfg, dtype)
    d["u_lo"] = np.full((N, m), cfg["u_lo"], dtype=dtype)
    d["u_hi"] = np.full((N, m), cfg["u_hi"], dtype=dtype)
    if "x_lo" in cfg:
        d["x_lo"], d["x_hi"] = cfg["x_lo"].astype(dtype), cfg["x_hi"].astype(dtype)
    return d


def rho_to_weights(rho, N, d, dtype=np.float64):
    """compute_Rr_Qr(dp=True) (isls/base.py:55-79): scalar / (d,d) / (N,d,d) -> (N,d,d)."""
    if rho is None:
        return None
    if np.isscalar(rho):
        return np.tile((rho * np.eye(d))[None], (N, 1, 1)).astype(dtype)
    rho = np.asarray(rho, dtype=dtype)
    if rho.ndim == 2:
        return np.tile(rho[None], (N, 1, 1))
    return rho.copy()


class OracleDriver:
    """Outer loop over oracle kernels (numpy).  Records a trace shaped like tests/golden 'o2_*'."""

    def __init__(self, kern, pa, rho_x=None, rho_u=None, project_x=False, project_u=True, relax=1.0,
                 dtype=np.float64, x_sets=None):
        """x_sets: an isls.projections.ConvexSets on the state rows (ISLS_PROJ_SETS) instead of the box pa['x_lo/hi']."""
        self.kern, self.pa, self.dtype = kern, pa, dtype
        self.x_sets = x_sets
        B, N, n, m = pa["B"], pa["N"], pa["n"], pa["m"]
        self.B, self.N, self.n, self.m = B, N, n, m
        self.Qr = rho_to_weights(rho_x, N, n, dtype) if project_x else None
        self.Rr = rho_to_weights(rho_u, N, m, dtype) if project_u else None
        self.wq = None if self.Qr is None else np.ascontiguousarray(self.Qr.sum(-1))   # (dx*dx)@Qr: row sums
        self.wr = None if self.Rr is None else np.ascontiguousarray(self.Rr.sum(-1))
        self.relax = relax
        z = lambda *s: np.zeros(s, dtype=dtype)   # noqa: E731
        self.xhat, self.uhat = pa["xhat"].copy(), pa["uhat"].copy()
        self.A, self.Bm = z(B, N, n, n), z(B, N, n, m)
        self.Cxx, self.Cuu, self.c0x, self.c0u = z(B, N, n, n), z(B, N, m, m), z(B, N, n), z(B, N, m)
        self.K, self.Quu, self.fac, self.Qux, self.k = z(B, N, m, n), z(B, N, m, m), z(B, N, m, m), z(B, N, m, n), z(B, N, m)
        self.xx, self.xu = z(B, N, n), z(B, N, m)
        self.zx = z(B, N, n) if project_x else None
        self.lx = z(B, N, n) if project_x else None
        self.zu = z(B, N, m) if project_u else None
        self.lu = z(B, N, m) if project_u else None
        self.cost, self.cost_new = z(B), z(B)
        self.best = np.zeros(B, dtype=np.int32)
        self.status = np.zeros(B, dtype=np.int32)
        self.res, self.res_prev = z(B, 2), z(B, 2)
        self.outer_active = np.ones(B, dtype=np.int32)
        self.admm_active = np.ones(B, dtype=np.int32)
        self.cost_log = [[] for _ in range(B)]
        # initial cost (nominal_values setter, isls/isls_base.py:80-85)
        self.kern.expand_quadratic(pa["Qtab"], pa["ztab"], pa["seq"], pa["u_std"], self.c0x, self.c0u,
                                   xhat=self.xhat, uhat=self.uhat, cost=self.cost, **self.cost_kw())
        for b in range(B):
            self.cost_log[b].append(float(self.cost[b]))

    def linearize_expand(self):
        pa = self.pa
        self.kern.linearize(pa["model"], pa["model_par"], self.xhat, self.uhat, self.A, self.Bm)
        self.kern.expand_quadratic(pa["Qtab"], pa["ztab"], pa["seq"], pa["u_std"], self.c0x, self.c0u,
                                   xhat=self.xhat, uhat=self.uhat, Cxx=self.Cxx, Cuu=self.Cuu,
                                   Qr=self.Qr, Rr=self.Rr, **self.cost_kw())

    def gain(self):
        self.kern.riccati_gain(self.A, self.Bm, self.Cxx, self.Cuu, self.K, self.Quu, self.fac, self.Qux,
                               status=self.status, active=self.admm_active)

    def ff(self):
        self.kern.riccati_ff(self.A, self.Bm, self.c0x, self.c0u, self.K, self.Quu, self.fac, self.Qux, self.k,
                             Qr=self.Qr, Rr=self.Rr, xhat=self.xhat, uhat=self.uhat, zx=self.zx, lx=self.lx,
                             zu=self.zu, lu=self.lu, active=self.admm_active)

    def rollout(self, L, flags=0, cost_all=None):
        pa = self.pa
        alphas = ALPHAS[:L].astype(self.dtype)
        self.kern.rollout_ls(pa["model"], pa["model_par"], self.K, self.k, self.xhat, self.uhat, alphas,
                             pa["Qtab"], pa["ztab"], pa["seq"], pa["u_std"], self.xx, self.xu,
                             best=self.best, cost_new=self.cost_new, cost_all=cost_all,
                             wq=self.wq, wr=self.wr, zx=self.zx, lx=self.lx, zu=self.zu, lu=self.lu,
                             cost_cur=self.cost, flags=flags, status=self.status, active=self.admm_active, **self.cost_kw())

    def cost_kw(self):
        """cost-model keywords (ISLS_COST_PHUBER problems carry pa['cost_model'], pa['cost_par'])."""
        if "cost_model" not in self.pa:
            return {}
        return dict(cost_model=self.pa["cost_model"], cost_par=self.pa["cost_par"])

    def set_args(self):
        """(x_sets descriptor, x_col0, x_work) keyword arguments of admm_args for the ConvexSets state constraint."""
        if self.x_sets is None:
            return {}
        if getattr(self, "_xs", None) is None:
            cs = self.x_sets
            self.x_work = np.zeros((self.B, self.N, self.n), dtype=self.dtype)
            wrap = lambda a: a if a.dtype.kind in "iu" else np.ascontiguousarray(a, dtype=self.dtype)     # noqa: E731
            self._xs = capi.Kernels.project_args_chain(self.x_work, self.x_work, cs.stages(), wrap=wrap)
        return dict(x_sets=self._xs, x_col0=self.x_sets.cols[0], x_work=self.x_work)

    def update(self, tol):
        pa = self.pa
        if self.x_sets is not None:
            self.kern.admm_update(self.xx, self.xu, self.res, zx=self.zx, lx=self.lx, zu=self.zu, lu=self.lu,
                                  u_lo=pa["u_lo"] if self.zu is not None else None,
                                  u_hi=pa["u_hi"] if self.zu is not None else None,
                                  relax=self.relax, tol_abs=tol, tol_rel=tol, res_prev=self.res_prev,
                                  active=self.admm_active, **self.set_args())
            return
        self.kern.admm_update(self.xx, self.xu, self.res, zx=self.zx, lx=self.lx, zu=self.zu, lu=self.lu,
                              x_lo=pa.get("x_lo") if self.zx is not None else None,
                              x_hi=pa.get("x_hi") if self.zx is not None else None,
                              u_lo=pa["u_lo"] if self.zu is not None else None,
                              u_hi=pa["u_hi"] if self.zu is not None else None,
                              relax=self.relax, tol_abs=tol, tol_rel=tol, res_prev=self.res_prev,
                              active=self.admm_active)

    def run_c(self, L, J, tol=0.0, log=None):
        """One outer iteration through the library's own driver entry point (`*_ilqr_admm_outer`):
        linearise + expand, then gain -> J x [ff -> rollout -> update] in one C call, then accept."""
        pa, K = self.pa, capi.Kernels
        alphas = ALPHAS[:L].astype(self.dtype)
        self.linearize_expand()
        gain = K.gain_args(self.A, self.Bm, self.Cxx, self.Cuu, self.K, self.Quu, self.fac, self.Qux,
                           status=self.status, active=self.admm_active)
        ff = K.ff_args(self.A, self.Bm, self.c0x, self.c0u, self.K, self.Quu, self.fac, self.Qux, self.k,
                       Qr=self.Qr, Rr=self.Rr, xhat=self.xhat, uhat=self.uhat, zx=self.zx, lx=self.lx,
                       zu=self.zu, lu=self.lu, active=self.admm_active)
        ro = K.rollout_args(pa["model"], pa["model_par"], self.K, self.k, self.xhat, self.uhat, alphas,
                            pa["Qtab"], pa["ztab"], pa["seq"], pa["u_std"], self.xx, self.xu, best=self.best,
                            cost_new=self.cost_new, wq=self.wq, wr=self.wr, zx=self.zx, lx=self.lx,
                            zu=self.zu, lu=self.lu, cost_cur=self.cost, status=self.status, active=self.admm_active,
                            **self.cost_kw())
        admm = K.admm_args(self.xx, self.xu, self.res, zx=self.zx, lx=self.lx, zu=self.zu, lu=self.lu,
                           x_lo=pa.get("x_lo") if self.zx is not None else None,
                           x_hi=pa.get("x_hi") if self.zx is not None else None,
                           u_lo=pa["u_lo"] if self.zu is not None else None,
                           u_hi=pa["u_hi"] if self.zu is not None else None,
                           relax=self.relax, tol_abs=tol, tol_rel=tol, res_prev=self.res_prev, active=self.admm_active)
        self._keep = (alphas,)
        sfx = "f64" if self.dtype == np.float64 else "f32"
        self.kern.outer(gain, ff, ro, admm, J, sfx, log=log, outer_active=self.outer_active)
        self.kern.accept_step(self.xx, self.xu, self.cost_new, self.xhat, self.uhat, self.cost,
                              outer_active=self.outer_active)

    def run(self, max_iter, L, J, tol):
        """Returns trace[outer] = dict(K, k[J], xx[J], xu[J], regx[J], regu[J], logs[J,B,2], cost[B], z.., n_inner[B])."""
        B = self.B
        trace = []
        for j in range(max_iter):
            if not self.outer_active.any():
                break
            act0 = self.outer_active.copy()
            prev_cost = self.cost.copy()
            self.linearize_expand()
            self.admm_active[:] = self.outer_active
            for arr in (self.lx, self.lu):
                if arr is not None:
                    arr[act0 == 1] = 0                      # lmb re-zeroed each outer iteration (isls.py:414-415,482)
            self.res_prev[:] = 1e6                          # admm.py:25-26
            self.gain()
            it = dict(K=None, k=[], xx=[], xu=[], regx=[], regu=[], logs=[], n_inner=np.zeros(B, dtype=np.int32),
                      active=act0)
            for i in range(J):
                if not self.admm_active.any():
                    break
                act = self.admm_active.copy()
                it["regx"].append(None if self.zx is None else (self.zx - self.lx).copy())
                it["regu"].append(None if self.zu is None else (self.zu - self.lu).copy())
                self.ff()
                self.rollout(L)
                self.update(tol)
                it["k"].append(self.k.copy()), it["xx"].append(self.xx.copy()), it["xu"].append(self.xu.copy())
                it["logs"].append(self.res.copy())
                it["n_inner"] += act
            it["K"] = self.K.copy()
            # nominal_values <- last x-step (isls.py:488); cost = plain cost of that trajectory
            a = act0 == 1
            self.xhat[a], self.uhat[a], self.cost[a] = self.xx[a], self.xu[a], self.cost_new[a]
            it["cost"] = self.cost.copy()
            it["zx"] = None if self.zx is None else self.zx.copy()
            it["zu"] = None if self.zu is None else self.zu.copy()
            it["lx"] = None if self.lx is None else self.lx.copy()
            it["lu"] = None if self.lu is None else self.lu.copy()
            trace.append(it)
            for b in range(B):
                if not a[b]:
                    continue
                self.cost_log[b].append(float(self.cost[b]))
                cl = self.cost_log[b]
                if abs(self.cost[b] - prev_cost[b]) < 1e-3:                                   # isls.py:493
                    self.outer_active[b] = 0
                elif len(cl[-8:-4]) and abs(np.mean(cl[-4:]) - np.mean(cl[-8:-4])) < 1e-3:    # isls.py:497
                    self.outer_active[b] = 0
        return trace


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def tassa_arrays(g, bsel, dtype=np.float64):
    """problem_arrays for the Tassa car-parking golden case (tests/golden/g8_tassa.npz): model ISLS_MODEL_TASSA, cost
    ISLS_COST_PHUBER, nominal = open-loop rollout of the recorded u0 from x0 (the notebook's get_trajectory_batch)."""
    bsel = list(bsel)
    N = int(g["N"])
    d = dict(B=len(bsel), N=N, n=4, m=2, model=capi.MODEL_TASSA, u_std=0.0,
             model_par=np.array([float(g["dt"]), float(g["dist"])], dtype=dtype),
             Qtab=np.zeros((1, 4, 4), dtype=dtype), ztab=np.zeros((1, 4), dtype=dtype), seq=np.zeros(N, dtype=np.int32),
             cost_model=capi.COST_PHUBER,
             cost_par=np.concatenate([g["par_cu"], g["par_cx"], g["par_px"], g["par_cf"], g["par_pf"]]).astype(dtype),
             xhat=np.ascontiguousarray(g["x_nom0"][bsel]).astype(dtype), uhat=np.ascontiguousarray(g["u0"][bsel]).astype(dtype),
             u_lo=np.tile(np.array([-0.5, -2.0], dtype=dtype), (N, 1)), u_hi=np.tile(np.array([0.5, 2.0], dtype=dtype), (N, 1)))
    return d


def outer_iteration_on_device(cfg, bsel, hip, oracle_kern, L, J, rho_u, relax=1.0, dtype="f64", outer_iters=1, scramble_best=None,
                              structured=False):
    """One outer DP-form iLQR-ADMM iteration through the library's own driver (`isls_ilqr_admm_outer_*`: gain pass with the
    first feed-forward pass inside, J x [ff -> rollout with the fused ADMM update]) on the device, and the same through the
    oracle's driver on the host; returns the worst relative error over K, k, the x-step, z, lambda and the residuals.
    dtype "f64" / "f32" selects isls_ilqr_admm_outer_f64 / _f32 (and the oracle of the same precision); outer_iters > 1 repeats
    the iteration (linearise + expand, driver call, accept) so that the gain pass also sees a moved nominal.  scramble_best (a
    seed): the `best` array the rollout reads as its PREDICTION of the winner is filled with random candidate indices first, so
    that wavefronts with mispredicted, correctly predicted and mixed winners all occur (recorded winner vs replay).
    structured: the gain pass and the feed-forward passes get the model hint isls.Engine gives them for this workload
    (isls_gain_args.lin_on / isls_ff_args.lin_on: the linearisation is a double integrator's): lean records, the Riccati passes
    on the model's structure -- against the same oracle run on the dense arrays."""
    import torch
    f = np.float64 if dtype == "f64" else np.float32
    o = OracleDriver(oracle_kern, problem_arrays(cfg, bsel, dtype=f), rho_u=rho_u, relax=relax, dtype=f)
    for _ in range(outer_iters):
        o.run_c(L, J)
    h = OracleDriver(oracle_kern, problem_arrays(cfg, bsel, dtype=f), rho_u=rho_u, relax=relax, dtype=f)
    for k in [k for k, v in vars(h).items() if isinstance(v, np.ndarray)]:
        setattr(h, k, torch.from_numpy(getattr(h, k)).cuda())
    h.pa = {k: (torch.from_numpy(v).cuda() if isinstance(v, np.ndarray) else v) for k, v in h.pa.items()}
    h.kern = hip
    B, N, n, m = h.B, h.N, h.n, h.m
    rec = torch.zeros(capi.ff_record_elems(B, N, n, m), dtype=torch.float64 if dtype == "f64" else torch.float32, device="cuda")
    pa, K = h.pa, capi.Kernels
    alphas = torch.from_numpy(ALPHAS[:L].astype(f)).cuda()
    lin = None
    if structured:
        from dual import DualKernels
        lin = DualKernels._lin_hint(pa["model"], np.asarray(problem_arrays(cfg, bsel, dtype=f)["model_par"]), pa["model_par"])
        assert lin is not None, "this workload has no model-structured form"
        h._lin_par = lin[1]                                    # the argument blocks hold its address
    gain = K.gain_args(h.A, h.Bm, h.Cxx, h.Cuu, h.K, None, None, None, status=h.status, active=h.admm_active, rec=rec, lin=lin)
    ff = K.ff_args(h.A, h.Bm, h.c0x, h.c0u, h.K, None, None, None, h.k, Rr=h.Rr[:1], xhat=h.xhat, uhat=h.uhat, zu=h.zu, lu=h.lu,
                   active=h.admm_active, rec=rec, lin=lin)
    ro = K.rollout_args(pa["model"], pa["model_par"], h.K, h.k, h.xhat, h.uhat, alphas, pa["Qtab"], pa["ztab"], pa["seq"],
                        pa["u_std"], h.xx, h.xu, best=h.best, cost_new=h.cost_new, wr=h.wr[:1], zu=h.zu, lu=h.lu, cost_cur=h.cost,
                        status=h.status, active=h.admm_active)
    admm = K.admm_args(h.xx, h.xu, h.res, zu=h.zu, lu=h.lu, u_lo=pa["u_lo"], u_hi=pa["u_hi"], relax=h.relax, tol_abs=0.0,
                       tol_rel=0.0, res_prev=h.res_prev, active=h.admm_active)
    for it_ in range(outer_iters):
        h.linearize_expand()
        if scramble_best is not None:
            h.best.copy_(torch.from_numpy(np.random.default_rng(scramble_best + it_).integers(0, L, size=B).astype(np.int32)))
        hip.outer(gain, ff, ro, admm, J, dtype, outer_active=h.outer_active)
        hip.accept_step(h.xx, h.xu, h.cost_new, h.xhat, h.uhat, h.cost, outer_active=h.outer_active)
    torch.cuda.synchronize()
    return max(rel_err(getattr(h, name).cpu().numpy(), getattr(o, name)) for name in ("K", "k", "xx", "xu", "zu", "lu", "res", "xhat", "uhat", "cost"))
