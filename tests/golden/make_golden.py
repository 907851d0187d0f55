#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (build container only).

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg PYTHONPATH=/root/reference \
        python3 tests/golden/make_golden.py

The reference (`/root/reference/isls`, pure python/numpy/scipy) is imported, never copied; only its
*outputs* on seeded inputs are stored (small .npz files).  Nothing here runs on the GPU box.

What the reference runs unmodified: all of `SLS`, `ADMM`, `projections`,
`iSLS.backward_pass_DP / rollout_DP / iterate_once_dp / solve(method='dp', get_Cs=...)`.
HEAD's `iSLS` quadratic-cost path has API drift (no `compute_cost`, `C`/`D` undefined, `ADMM(threshold=)`),
so `RefISLS` below is the documented compatibility shim: a subclass that binds the missing names to
the reference's OWN implementations (`SLSBase.compute_cost`, `Sw`, `Su`) -- no numerics of ours.

"O2" = DP-form iLQR-ADMM composed ONLY of reference functions (regularised Cts/cts ->
`backward_pass_DP` -> `rollout_DP` -> `cost_function` + AL terms -> argmin, as `f_argmin` of the
reference's `ADMM()`), with the outer-loop semantics of `iSLS.ilqr_admm` (isls/isls.py:379-501).
The generator asserts that O2 reproduces the reference's own batch-form `ilqr_admm` (shimmed).
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))

import isls as ref                                    # noqa: E402  (the REFERENCE package)
import isls.isls as ref_isls_mod                      # noqa: E402
from isls.admm import ADMM as REF_ADMM                # noqa: E402
from isls.sls_base import SLSBase as RefSLSBase       # noqa: E402
refproj = sys.modules["isls.projections"]             # `isls.projections` the attribute is shadowed by a dict

assert ref.__file__.startswith("/root/reference/"), ref.__file__

_spec = importlib.util.spec_from_file_location(
    "isls_problems", os.path.join(REPO, "ilqr-admm_amd", "isls_problems.py"))
P = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(P)


class RefISLS(ref.iSLS):
    """Compatibility shim: bind names HEAD lost to the reference's own code."""
    compute_cost = RefSLSBase.compute_cost
    C = property(lambda self: self.Sw)
    D = property(lambda self: self.Su)

    def set_cost_variables(self, zs, Qs, seq, u_std):
        return self.set_quadratic_cost(zs, Qs, seq, u_std)


def _admm_threshold_shim(*a, threshold=None, **kw):
    if threshold is not None:
        kw["tol"] = threshold
    return REF_ADMM(*a, **kw)


ref_isls_mod.ADMM = _admm_threshold_shim             # `ilqr_admm` passes threshold= (isls.py:485)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path)/1024:.1f} KiB, keys={len(arrays)}")


# ---------------------------------------------------------------------------------------------
# G1/G2: LQT on the 1-D double integrator (unmodified SLS)
# ---------------------------------------------------------------------------------------------
def gen_di1d():
    out = {}
    for tag, N in (("n100", 100), ("n50", 50)):
        c = P.config1(N)
        sls = ref.SLS(2, 1, N)
        sls.AB = [c["A"], c["B"]]
        sls.set_quadratic_cost(c["zs"], c["Qs"], c["seq"], c["u_std"])
        x_opt, u_opt = sls.solve(c["x0"], method="batch")
        out[f"{tag}_batch_x"] = x_opt
        out[f"{tag}_batch_u"] = u_opt
        if N == 100:   # pins recorded in the notebook (control bounds.ipynb:154-156)
            assert np.max(u_opt) == 6.06051888764695, np.max(u_opt)
            assert x_opt[-1, 0] == 0.9999876316133441
        K, k = sls.solve_dp()
        out[f"{tag}_dp_K"], out[f"{tag}_dp_k"] = K, k
        Qr, Rr = sls.compute_Rr_Qr(rho_x=None, rho_u=c["rho_u"], dp=True)
        rng = np.random.default_rng(11)
        ur = rng.standard_normal(N)
        K2, k2, Quu, Quu_inv, Qux = sls.solve_dp(Rr=Rr, Qr=Qr, xr=np.zeros(2 * N), ur=ur, return_Qs=True)
        out[f"{tag}_reg_ur"] = ur
        out[f"{tag}_reg_K"], out[f"{tag}_reg_k"] = K2, k2
        out[f"{tag}_reg_Quu"], out[f"{tag}_reg_Quu_inv"], out[f"{tag}_reg_Qux"] = Quu, Quu_inv, Qux
        ur2 = rng.standard_normal(N)
        out[f"{tag}_ff_ur"] = ur2
        out[f"{tag}_ff_k"] = sls.solve_dp_ff(K2, Quu, Qux, Quu_inv, Qr=Qr, Rr=Rr, ur=ur2, xr=np.zeros(2 * N))
        # state-regularised variant (Qr) as well
        Qr3, Rr3 = sls.compute_Rr_Qr(rho_x=0.5, rho_u=c["rho_u"], dp=True)
        xr3 = rng.standard_normal(2 * N)
        K3, k3, Quu3, Quu_inv3, Qux3 = sls.solve_dp(Rr=Rr3, Qr=Qr3, xr=xr3, ur=ur, return_Qs=True)
        out[f"{tag}_regx_xr"] = xr3
        out[f"{tag}_regx_K"], out[f"{tag}_regx_k"] = K3, k3
        proj = lambda u: refproj.project_bound(u, c["u_lo"], c["u_hi"])   # noqa: E731
        xa, ua, Ka, ka, logs = sls.ADMM_LQT_DP(c["x0"], project_u=proj, max_iter=500, rho_u=c["rho_u"],
                                               tol=c["tol"], verbose=False, log=True)
        out[f"{tag}_admm_dp_x"], out[f"{tag}_admm_dp_u"] = xa, ua
        out[f"{tag}_admm_dp_K"], out[f"{tag}_admm_dp_k"] = Ka, ka
        out[f"{tag}_admm_dp_logs"] = np.stack(logs)
        out[f"{tag}_admm_dp_cost"] = np.array(sls.compute_cost(xa, ua))
        # fixed-iteration variant for kernel-level traces (no early stop: tol=0)
        xa, ua, Ka, ka, logs = sls.ADMM_LQT_DP(c["x0"], project_u=proj, max_iter=8, rho_u=c["rho_u"],
                                               tol=0.0, verbose=False, log=True)
        out[f"{tag}_admm8_x"], out[f"{tag}_admm8_u"], out[f"{tag}_admm8_k"] = xa, ua, ka
        out[f"{tag}_admm8_logs"] = np.stack(logs)
        xb, ub, logb = sls.ADMM_LQT_Batch(x0=c["x0"], project_u=proj, max_iter=100, rho_u=1e-2, tol=1e-4,
                                          verbose=False, log=True)
        out[f"{tag}_admm_batch_u"] = ub
        out[f"{tag}_admm_batch_logs"] = np.stack(logb)
        if N == 100:   # control bounds.ipynb:204-207
            assert len(logb) == 20 and np.max(ub) == 5.000018035934772, (len(logb), np.max(ub))
    save("g1_di1d_lqt.npz", **out)


# ---------------------------------------------------------------------------------------------
# O2: DP-form iLQR-ADMM from reference functions
# ---------------------------------------------------------------------------------------------
def make_ref_isls(cfg, b):
    """Reference iSLS object for trajectory b of a batched config."""
    n, m, N = cfg["n"], cfg["m"], cfg["N"]
    f, get_AB = P.model_callbacks(cfg)
    obj = RefISLS(n, m, N)
    obj.forward_model = f
    zs = cfg["zs"][b] if cfg["zs"].ndim == 3 else cfg["zs"]
    obj.set_cost_variables(zs, cfg["Qs"], cfg["seq"], cfg["u_std"])
    x_nom, u_nom = P.initial_nominal(cfg, b)
    obj.reset()
    obj.nominal_values = x_nom, u_nom
    return obj, get_AB


def expansions(obj, Qr, Rr, rx, ru):
    """Regularised quadratic-cost expansion in delta coordinates."""
    n, m, N = obj.x_dim, obj.u_dim, obj.N
    Cts = np.zeros((N, n + m, n + m))
    cts = np.zeros((N, n + m))
    user = getattr(obj, "_gen_get_Cs", None)                 # non-quadratic cost: the user's get_Cs callback (isls.py:102)
    if user is not None:
        cts, Cts = user(obj.x_nom, obj.u_nom)
        cts, Cts = cts.copy(), Cts.copy()
    for t in range(N):
        if user is None:
            Q = obj.Qs[obj.seq[t]]
            Cts[t, :n, :n] = 2 * Q
            Cts[t, n:, n:] = 2 * obj.Rt
            cts[t, :n] = 2 * Q.dot(obj.x_nom[t] - obj.zs[obj.seq[t]])
            cts[t, n:] = 2 * obj.Rt.dot(obj.u_nom[t])
        if Qr is not None:
            Cts[t, :n, :n] += 2 * Qr[t]
            cts[t, :n] += 2 * Qr[t].dot(obj.x_nom[t] - rx[t])
        if Rr is not None:
            Cts[t, n:, n:] += 2 * Rr[t]
            cts[t, n:] += 2 * Rr[t].dot(obj.u_nom[t] - ru[t])
    return Cts, cts


def o2_ilqr_admm(obj, get_AB, project_x, project_u, rho_x, rho_u, max_iter, L, J, relax, tol, trace):
    """Outer loop with the semantics of iSLS.ilqr_admm (isls/isls.py:420-499), DP inner solve."""
    n, m, N = obj.x_dim, obj.u_dim, obj.N
    Qr, Rr = obj.compute_Rr_Qr(rho_x=rho_x, rho_u=rho_u, dp=True)
    if not project_x:
        Qr = None
    if not project_u:
        Rr = None
    import scipy.linalg
    Qr_bd = scipy.linalg.block_diag(*Qr) if Qr is not None else None
    Rr_bd = scipy.linalg.block_diag(*Rr) if Rr is not None else None
    alphas = obj.alphas[:L]
    z_x_init = z_u_init = None
    n_outer = 0
    for j in range(max_iter):
        prev_cost = np.copy(obj.cost)
        obj.A, obj.B = get_AB(obj.x_nom, obj.u_nom)
        inner = dict(K=None, k=[], xx=[], xu=[], regx=[], regu=[], cost_aug=[], ind=[])

        def f_argmin(x, u):
            rx = x.reshape(N, n) if x is not None else None
            ru = u.reshape(N, m) if u is not None else None
            Cts, cts = expansions(obj, Qr, Rr, rx, ru)
            K, k = obj.backward_pass_DP(Cts=Cts, cts=cts)
            k_new = k[None] * alphas[:, None, None]
            x_noms, u_noms = obj.rollout_DP(K, k_new)
            costs = obj.cost_function(x_noms, u_noms)
            costs = np.atleast_1d(costs).astype(float)
            if Qr is not None:
                dx = x_noms.reshape(-1, N * n) - x
                costs += np.sum(dx * dx @ Qr_bd, axis=-1)
            if Rr is not None:
                du = u_noms.reshape(-1, N * m) - u
                costs += np.sum(du * du @ Rr_bd, axis=-1)
            ind = np.argmin(costs)
            inner["K"] = K
            inner["k"].append(k)
            inner["xx"].append(x_noms[ind].copy())
            inner["xu"].append(u_noms[ind].copy())
            inner["regx"].append(np.zeros((N, n)) if rx is None else rx.copy())
            inner["regu"].append(np.zeros((N, m)) if ru is None else ru.copy())
            inner["cost_aug"].append(costs.copy())
            inner["ind"].append(ind)
            return x_noms[ind].flatten(), u_noms[ind].flatten()

        admm = REF_ADMM(n * N, m * N, f_argmin, project_x=project_x, project_u=project_u,
                        z_x_init=z_x_init, z_u_init=z_u_init, lmb_x_init=None, lmb_u_init=None,
                        return_lmb=1, alpha=relax, max_iter=J, tol=tol, verbose=False, log=True)
        obj.nominal_values = admm[0].reshape(N, -1), admm[1].reshape(N, -1)
        lmb_x, lmb_u, z_x, z_u, logs = admm[2], admm[3], admm[4], admm[5], admm[6]
        z_x_init, z_u_init = z_x, z_u
        n_outer += 1
        trace.append(dict(inner=inner, logs=np.stack(logs), cost=float(obj.cost),
                          z_x=None if z_x is None else z_x.copy(), z_u=None if z_u is None else z_u.copy(),
                          lmb_x=None if lmb_x is None else lmb_x.copy(),
                          lmb_u=None if lmb_u is None else lmb_u.copy(),
                          x_nom=obj.x_nom.copy(), u_nom=obj.u_nom.copy()))
        if np.abs(obj.cost - prev_cost) < 1e-3:
            break
        if np.abs(np.mean(obj.cost_log[-4:]) - np.mean(obj.cost_log[-8:-4])) < 1e-3:
            break
    return n_outer


def pack_trace(prefix, traces, n, m, N, J, out):
    """traces: list over trajectories of list over outer its. Ragged inner counts padded with NaN."""
    Bt = len(traces)
    n_outer = max(len(t) for t in traces)
    shp = (Bt, n_outer)
    K = np.full(shp + (N, m, n), np.nan)
    k = np.full(shp + (J, N, m), np.nan)
    xx = np.full(shp + (J, N, n), np.nan)
    xu = np.full(shp + (J, N, m), np.nan)
    regx = np.full(shp + (J, N, n), np.nan)
    regu = np.full(shp + (J, N, m), np.nan)
    logs = np.full(shp + (J, 2), np.nan)
    cost = np.full(shp, np.nan)
    zx = np.full(shp + (N, n), np.nan)
    zu = np.full(shp + (N, m), np.nan)
    lx = np.full(shp + (N, n), np.nan)
    lu = np.full(shp + (N, m), np.nan)
    n_inner = np.zeros(shp, dtype=np.int32)
    outer_count = np.array([len(t) for t in traces], dtype=np.int32)
    for b, tr in enumerate(traces):
        for o, it in enumerate(tr):
            ji = len(it["inner"]["k"])
            n_inner[b, o] = ji
            K[b, o] = it["inner"]["K"]
            k[b, o, :ji] = np.stack(it["inner"]["k"])
            xx[b, o, :ji] = np.stack(it["inner"]["xx"])
            xu[b, o, :ji] = np.stack(it["inner"]["xu"])
            regx[b, o, :ji] = np.stack(it["inner"]["regx"])
            regu[b, o, :ji] = np.stack(it["inner"]["regu"])
            logs[b, o, :ji] = it["logs"]
            cost[b, o] = it["cost"]
            if it["z_x"] is not None:
                zx[b, o] = it["z_x"].reshape(N, n)
                lx[b, o] = it["lmb_x"].reshape(N, n)
            if it["z_u"] is not None:
                zu[b, o] = it["z_u"].reshape(N, m)
                lu[b, o] = it["lmb_u"].reshape(N, m)
    out.update({f"{prefix}_K": K, f"{prefix}_k": k, f"{prefix}_xx": xx, f"{prefix}_xu": xu,
                f"{prefix}_regx": regx, f"{prefix}_regu": regu, f"{prefix}_logs": logs,
                f"{prefix}_cost": cost, f"{prefix}_zx": zx, f"{prefix}_zu": zu, f"{prefix}_lx": lx,
                f"{prefix}_lu": lu, f"{prefix}_n_inner": n_inner, f"{prefix}_n_outer": outer_count})


def trace_sensitivity(cfg, bsel, run_o2, keys=("K", "k", "xx", "xu", "logs", "cost")):
    """Conditioning probe: re-run the REFERENCE O2 trace with the initial controls perturbed by one
    part in 1e15 and record, per quantity, max|delta| / max(1,|ref|_max).  A restatement that differs
    from the reference only by floating-point summation order cannot be expected to agree better
    than this; the parity tests use max(1e-10, 10 x sensitivity) as their tolerance."""
    base, pert = {}, {}
    for tag, store, eps in (("base", base, 0.0), ("pert", pert, 1e-15)):
        c = dict(cfg)
        c["u0"] = cfg["u0"] * (1.0 + eps) + eps
        traces = [run_o2(c, b) for b in bsel]
        pack_trace("s", traces, c["n"], c["m"], c["N"], max(len(t["inner"]["k"]) for tr in traces for t in tr), store)
    sens = {}
    for k in keys:
        a, b = base[f"s_{k}"], pert[f"s_{k}"]
        ok = ~(np.isnan(a) | np.isnan(b))
        sens[k] = float(np.max(np.abs(a[ok] - b[ok])) / max(1.0, float(np.max(np.abs(a[ok])))))
    return sens


def box_projectors(cfg):
    N, n, m = cfg["N"], cfg["n"], cfg["m"]
    proj_u = lambda u: refproj.project_bound(u, cfg["u_lo"], cfg["u_hi"])      # noqa: E731
    proj_x = False
    if "x_lo" in cfg:
        lo, hi = cfg["x_lo"].reshape(-1), cfg["x_hi"].reshape(-1)
        proj_x = lambda x: refproj.project_bound(x, lo, hi)                    # noqa: E731
    return proj_x, proj_u


def gen_di3d():
    out = {}
    Bt, L, J = 2, 20, 5
    cfg = P.config2(batch=4, N=100, seed=0)
    n, m, N = cfg["n"], cfg["m"], cfg["N"]
    out["cfg_batch"], out["cfg_seed"] = np.array(4), np.array(0)
    # --- kernel-level vectors at the initial nominal -----------------------------------------
    Kq, kq, Kc, kc, costs, xsel, usel, it_once = [], [], [], [], [], [], [], []
    rxs, rus = [], []
    rng = np.random.default_rng(5)
    for b in range(Bt):
        obj, get_AB = make_ref_isls(cfg, b)
        obj.A, obj.B = get_AB(obj.x_nom, obj.u_nom)
        K, k = obj.backward_pass_DP()                      # quadratic branch (Cts=None)
        Kq.append(K), kq.append(k)
        rx = obj.x_nom + 0.1 * rng.standard_normal((N, n))
        ru = obj.u_nom + 0.1 * rng.standard_normal((N, m))
        Qr, Rr = obj.compute_Rr_Qr(rho_x=0.3, rho_u=1e-2, dp=True)
        Cts, cts = expansions(obj, Qr, Rr, rx, ru)
        K2, k2 = obj.backward_pass_DP(Cts=Cts, cts=cts)    # general branch, regularised
        Kc.append(K2), kc.append(k2), rxs.append(rx), rus.append(ru)
        k_new = k[None] * obj.alphas[:L, None, None]
        x_noms, u_noms = obj.rollout_DP(K, k_new)
        costs.append(obj.cost_function(x_noms, u_noms))
        xsel.append(x_noms[[0, 7, 19]]), usel.append(u_noms[[0, 7, 19]])
        ok, _, _ = obj.iterate_once_dp(max_line_search=L)
        it_once.append(np.concatenate([[float(ok), obj.cost], obj.x_nom.ravel(), obj.u_nom.ravel()]))
    out.update(bp_quad_K=np.stack(Kq), bp_quad_k=np.stack(kq), bp_reg_K=np.stack(Kc), bp_reg_k=np.stack(kc),
               bp_reg_rx=np.stack(rxs), bp_reg_ru=np.stack(rus), ro_costs=np.stack(costs),
               ro_x_sel=np.stack(xsel), ro_u_sel=np.stack(usel), iter_once=np.stack(it_once))
    # --- unconstrained iLQR (iterate_once_dp loop with the stop rules of isls.py:125-132) -----
    logs = []
    for b in range(Bt):
        obj, get_AB = make_ref_isls(cfg, b)
        for i in range(10):
            obj.A, obj.B = get_AB(obj.x_nom, obj.u_nom)
            ok, _, _ = obj.iterate_once_dp(max_line_search=L)
            if np.abs(np.diff(obj.cost_log[-2:])) < 1e-5 or not ok:
                break
        logs.append(np.pad(np.array(obj.cost_log, dtype=float), (0, 12 - len(obj.cost_log)), constant_values=np.nan))
    out["ilqr_cost_log"] = np.stack(logs)
    # --- O2 traces ----------------------------------------------------------------------------
    proj_x, proj_u = box_projectors(cfg)
    traces = []
    for b in range(Bt):
        obj, get_AB = make_ref_isls(cfg, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, False, proj_u, None, cfg["rho_u"], max_iter=3, L=L, J=J,
                     relax=cfg["relax"], tol=0.0, trace=tr)
        traces.append(tr)
    pack_trace("o2", traces, n, m, N, J, out)

    def run_di(c, b):
        obj, get_AB = make_ref_isls(c, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, False, proj_u, None, c["rho_u"], max_iter=3, L=L, J=J,
                     relax=c["relax"], tol=0.0, trace=tr)
        return tr
    sens = trace_sensitivity(cfg, [0, 1], run_di)
    print("di3d O2 sensitivity to a 1e-15 input perturbation:", sens)
    out["o2_sens_keys"] = np.array(list(sens.keys()))
    out["o2_sens"] = np.array(list(sens.values()))
    # natural stopping (tol=1e-3) -- exercises both ADMM stop rules and the outer stop rules
    traces = []
    for b in range(Bt):
        obj, get_AB = make_ref_isls(cfg, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, False, proj_u, None, cfg["rho_u"], max_iter=8, L=L, J=10,
                     relax=cfg["relax"], tol=1e-3, trace=tr)
        traces.append(tr)
    pack_trace("o2stop", traces, n, m, N, 10, out)
    # relaxation != 1 and a state box as well (rho_x scalar)
    cfgx = dict(cfg)
    cfgx["x_lo"] = np.full((N, n), -np.inf)
    cfgx["x_hi"] = np.full((N, n), np.inf)
    cfgx["x_lo"][:, 3:6], cfgx["x_hi"][:, 3:6] = -1.2, 1.2
    proj_x2, _ = box_projectors(cfgx)
    traces = []
    for b in range(Bt):
        obj, get_AB = make_ref_isls(cfgx, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, proj_x2, proj_u, 0.05, cfg["rho_u"], max_iter=2, L=L, J=4,
                     relax=1.5, tol=0.0, trace=tr)
        traces.append(tr)
    pack_trace("o2x", traces, n, m, N, 4, out)
    out["o2x_x_lo"], out["o2x_x_hi"] = cfgx["x_lo"], cfgx["x_hi"]

    # --- cross-check O2 against the reference's own batch-form ilqr_admm (shimmed) -------------
    obj, get_AB = make_ref_isls(cfg, 0)
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        logs_b = obj.ilqr_admm(get_AB, project_u=proj_u, max_iter=1, max_line_search_iter=L, max_admm_iter=J,
                               rho_u=cfg["rho_u"], alpha=1.0, tol=0.0, log=True)
    o2logs = out["o2_logs"][0, 0]
    err_logs = np.max(np.abs(np.stack(logs_b) - o2logs))
    err_x = np.max(np.abs(obj.x_nom - out["o2_xx"][0, 0, J - 1]))
    err_u = np.max(np.abs(obj.u_nom[:-1] - out["o2_xu"][0, 0, J - 1][:-1]))
    print(f"O2 vs reference batch-form ilqr_admm: logs {err_logs:.2e}, x {err_x:.2e}, u[:-1] {err_u:.2e}")
    assert err_logs < 1e-9 and err_x < 1e-9 and err_u < 1e-8
    out["xcheck_batchform"] = np.array([err_logs, err_x, err_u])
    save("g3_di3d.npz", **out)


def gen_arm():
    out = {}
    cfg = P.config3(batch=2, N=100, seed=0)
    n, m, N = cfg["n"], cfg["m"], cfg["N"]
    # --- pin: Task 2 of the robust notebook (recorded initial/converged costs) -----------------
    cfg2 = dict(cfg)
    target = np.array([0, 0, 0, 0, 0, 0, 1.5, 2.0, 0.0])
    Qf = np.diag([0, 0, 0, 1e3, 1e3, 1e3, 1e3, 1e3, 0.0])
    cfg2["zs"], cfg2["Qs"], cfg2["seq"] = P.via_point_cost(n, N, target, Qf)
    cfg2["u0"] = np.zeros_like(cfg["u0"])
    obj, get_AB = make_ref_isls(cfg2, 0)
    assert obj.cost == 6775.068343357641, obj.cost          # robust notebook :351
    for i in range(30):
        obj.A, obj.B = get_AB(obj.x_nom, obj.u_nom)
        ok, _, _ = obj.iterate_once_dp(max_line_search=20)
        if np.abs(np.diff(obj.cost_log[-2:])) < 1e-5 or not ok:
            break
    print("arm task2 cost log", obj.cost_log)
    assert abs(obj.cost - 0.1180803005667605) < 1e-9
    out["task2_cost_log"] = np.array(obj.cost_log, dtype=float)
    # --- Task 1: kernel-level + iLQR + O2 with the notebook constraint set ----------------------
    L, J = cfg["max_line_search"], cfg["max_admm_iter"]
    Kq, kq, AA, BB, logs = [], [], [], [], []
    for b in range(2):
        obj, get_AB = make_ref_isls(cfg, b)
        A, B = get_AB(obj.x_nom, obj.u_nom)
        obj.A, obj.B = A, B
        AA.append(A.copy()), BB.append(B.copy())
        K, k = obj.backward_pass_DP()
        Kq.append(K), kq.append(k)
        for i in range(30):
            obj.A, obj.B = get_AB(obj.x_nom, obj.u_nom)
            ok, _, _ = obj.iterate_once_dp(max_line_search=20)
            if np.abs(np.diff(obj.cost_log[-2:])) < 1e-5 or not ok:
                break
        logs.append(np.pad(np.array(obj.cost_log, dtype=float), (0, 32 - len(obj.cost_log)), constant_values=np.nan))
    out.update(lin_A=np.stack(AA), lin_B=np.stack(BB), bp_quad_K=np.stack(Kq), bp_quad_k=np.stack(kq),
               ilqr_cost_log=np.stack(logs))
    proj_x, proj_u = box_projectors(cfg)
    traces = []
    for b in range(2):
        obj, get_AB = make_ref_isls(cfg, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, proj_x, proj_u, cfg["rho_x"], cfg["rho_u"], max_iter=3, L=L, J=J,
                     relax=1.0, tol=0.0, trace=tr)
        traces.append(tr)
    pack_trace("o2", traces, n, m, N, J, out)

    def run_arm(c, b):
        obj, get_AB = make_ref_isls(c, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, proj_x, proj_u, c["rho_x"], c["rho_u"], max_iter=3, L=L, J=J,
                     relax=1.0, tol=0.0, trace=tr)
        return tr
    sens = trace_sensitivity(cfg, [0, 1], run_arm)
    print("arm O2 sensitivity to a 1e-15 input perturbation:", sens)
    out["o2_sens_keys"] = np.array(list(sens.keys()))
    out["o2_sens"] = np.array(list(sens.values()))
    # the notebook call itself (threshold=1e-4, natural stop) on trajectory 0: final cost only
    obj, get_AB = make_ref_isls(cfg, 0)
    tr = []
    n_outer = o2_ilqr_admm(obj, get_AB, proj_x, proj_u, cfg["rho_x"], cfg["rho_u"], max_iter=20, L=L, J=J,
                           relax=1.0, tol=1e-4, trace=tr)
    out["o2_notebook_cost_log"] = np.array(obj.cost_log, dtype=float)
    out["o2_notebook_n_outer"] = np.array(n_outer)
    print("arm O2 notebook-call cost log:", obj.cost_log)
    save("g4_arm3r.npz", **out)


def ref_keepout_projection(N, d):
    """State constraint of the car notebook (`Iterative LQR with state constraints`, cell 18) rebuilt from the reference's
    primitives: stay outside two rectangles (2 x 1 plus a 0.5 margin, rotated by -45 degrees, centred at (-7,-3) and
    (-3,-7)); each rectangle is a `project_square_batch` in a scaled, rotated frame and the two are intersected by
    `project_set_convex(rho=10, max_iter=15, threshold=1e-3)` over the N state rows.  Returns (project_x, rho_x)."""
    centres = ((-7.0, -3.0), (-3.0, -7.0))
    size = np.array([2.0, 1.0]) + 0.5
    ang = -np.pi / 4
    rot = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    frame = np.diag(size[0] / size) @ rot.T                     # rectangle -> square of half-width size[0]/2
    frame_inv = np.linalg.inv(frame)

    def keep_out(centre):
        c = np.asarray(centre)

        def proj(rows):
            out = rows.reshape(N, d).copy()
            local = (out[:, :2] - c) @ frame.T
            out[:, :2] = refproj.project_square_batch(local, size[0] / 2, 1e5) @ frame_inv.T + c
            return out
        return proj
    parts = [keep_out(c) for c in centres]

    def project_x(flat):
        return refproj.project_set_convex(flat.reshape(N, d).copy(), [np.eye(d)] * 2, [np.zeros(d)] * 2, parts, rho=1e1,
                                          max_iter=15, verbose=0, threshold=1e-3).flatten()
    rho_x = np.zeros((N, d, d))
    rho_x[:, 0, 0] = rho_x[:, 1, 1] = 1e-1
    return project_x, rho_x


def gen_car():
    out = {}
    cfg = P.config4(batch=2, N=200, seed=0)
    n, m, N = cfg["n"], cfg["m"], cfg["N"]
    L, J = 20, cfg["max_admm_iter"]
    AA, BB, Kq, kq = [], [], [], []
    for b in range(2):
        obj, get_AB = make_ref_isls(cfg, b)
        A, B = get_AB(obj.x_nom, obj.u_nom)
        obj.A, obj.B = A, B
        AA.append(A.copy()), BB.append(B.copy())
        K, k = obj.backward_pass_DP()
        Kq.append(K), kq.append(k)
    out.update(lin_A=np.stack(AA), lin_B=np.stack(BB), bp_quad_K=np.stack(Kq), bp_quad_k=np.stack(kq))
    proj_x, proj_u = box_projectors(cfg)
    traces = []
    for b in range(2):
        obj, get_AB = make_ref_isls(cfg, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, proj_x, proj_u, cfg["rho_x"], cfg["rho_u"], max_iter=3, L=L, J=J,
                     relax=1.0, tol=0.0, trace=tr)
        traces.append(tr)
    pack_trace("o2", traces, n, m, N, J, out)

    def run_car(c, b):
        obj, get_AB = make_ref_isls(c, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, proj_x, proj_u, c["rho_x"], c["rho_u"], max_iter=3, L=L, J=J,
                     relax=1.0, tol=0.0, trace=tr)
        return tr
    sens = trace_sensitivity(cfg, [0, 1], run_car)
    print("car O2 sensitivity to a 1e-15 input perturbation:", sens)
    out["o2_sens_keys"] = np.array(list(sens.keys()))
    out["o2_sens"] = np.array(list(sens.values()))
    # O2 with the notebook's STATE constraint (state constraints.ipynb cells 18-20): two rotated keep-out rectangles
    # intersected by project_set_convex (rho=10, 15 iterations, 1e-3), rho_x = 0.1 on the two positions, no control
    # constraint; the closures below are the notebook's cell 18 written with the reference's own functions
    project_state, rho_k = ref_keepout_projection(N, n)
    traces = []
    for b in range(2):
        obj, get_AB = make_ref_isls(cfg, b)
        tr = []
        o2_ilqr_admm(obj, get_AB, project_state, None, rho_k, None, max_iter=3, L=L, J=10, relax=1.0, tol=0.0, trace=tr)
        traces.append(tr)
    pack_trace("o2k", traces, n, m, N, 10, out)

    # pin: the notebook's own initial cost (state constraints.ipynb:214), x0=[0,-2,pi/2,0], u0 = 0
    c = dict(cfg)
    c["x0"] = np.array([[0.0, -2.0, np.pi / 2, 0.0]])
    c["u0"] = np.zeros((1, 500, 2))
    c["N"], c["dt"] = 500, 15.0 / 500
    c["zs"], c["Qs"], c["seq"] = P.via_point_cost(4, 500, [-5.0, -5.0, np.pi / 4, 0.0], 1e2 * np.eye(4))
    obj, _ = make_ref_isls(c, 0)
    print("car notebook initial cost", obj.cost)
    out["notebook_initial_cost"] = np.array(obj.cost)
    save("g5_car.npz", **out)


def gen_projections():
    rng = np.random.default_rng(3)
    out = {}
    x = rng.standard_normal((64, 5)) * 3
    out["bound_in"], out["bound_out"] = x, refproj.project_bound(x, -1.5, 2.0)
    zt = rng.standard_normal((200, 4)) * 2
    zt[:10, -1] = -np.abs(zt[:10, -1]) * 5            # deep inside the polar cone
    zt[10:20, :-1] *= 0.01                            # well inside the cone
    out["soc_in"], out["soc_out"] = zt, refproj.project_soc_unit(zt.copy())
    sq = rng.standard_normal((100, 2)) * 2
    out["square_in"], out["square_out"] = sq, refproj.project_square_batch(sq.copy(), 1.0, 2.5)
    qd = rng.standard_normal((100, 3)) * 2
    out["quad_in"], out["quad_out"] = qd, refproj.project_quadratic_batch(qd.copy(), 0.5, 3.0)
    a = rng.standard_normal((50, 3))
    xl = rng.standard_normal((50, 3)) * 3
    out["lin_in"], out["lin_a"] = xl, a
    out["lin_out"] = refproj.project_linear_batch(xl.copy(), a, -0.5, 1.0)
    # set intersection of two unit-SOC images (the chance-constraint rows)
    from scipy.stats import norm
    psi = norm.ppf(0.95)
    Au = np.diag(np.sqrt([0.0, 0.01]))
    mu = np.array([1.0, 0.0])
    A_ = [np.concatenate([Au, (-mu / psi)[None]], 0), np.concatenate([Au, (mu / psi)[None]], 0)]
    b_ = [np.append(np.zeros(2), 5.0 / psi), np.append(np.zeros(2), 5.0 / psi)]
    y = rng.standard_normal((40, 2)) * np.array([6.0, 30.0])
    out["setcvx_in"] = y
    out["setcvx_A0"], out["setcvx_A1"], out["setcvx_b0"], out["setcvx_b1"] = A_[0], A_[1], b_[0], b_[1]
    out["setcvx_out"] = refproj.project_set_convex(y.copy(), A_, b_, projections=[refproj.project_soc_unit] * 2,
                                                   rho=1e1, max_iter=100, threshold=1e-3)
    # keep-out rectangles of the car notebook on [N,4] state rows
    proj_state, _ = ref_keepout_projection(200, 4)
    xk = rng.standard_normal((200, 4)) * np.array([3.0, 3.0, 1.0, 1.0]) + np.array([-5.0, -5.0, 0.0, 0.0])
    out["keepout_in"], out["keepout_out"] = xk, proj_state(xk.reshape(-1).copy()).reshape(200, 4)
    # the obstacle notebook's state projection (cell 12): project_set_convex over spherical keep-out shells, then Dykstra
    # (appended after the draws above so that every earlier vector keeps its value)
    centres, radii = [np.array([0.5, 0.5]), np.array([0.5, 0.2])], [0.1, 0.15]
    lowers = [0.5 * (1.1 * r) ** 2 for r in radii]
    shells = [lambda x, lo=lo, c=c: refproj.project_quadratic(x - c, lo, 1e2) + c for lo, c in zip(lowers, centres)]
    pts = rng.uniform(0.0, 1.0, size=(60, 2))
    out["shell_in"], out["shell_centres"], out["shell_lowers"] = pts, np.stack(centres), np.array(lowers)
    out["shell_admm_out"] = refproj.project_set_convex(pts.copy(), [np.eye(2)] * 2, [np.zeros(2)] * 2, shells, max_iter=5,
                                                       verbose=0, threshold=1e-2)
    out["shell_dykstra_out"] = refproj.project_set_convex_dykstra(out["shell_admm_out"].copy(), shells, max_iter=50, verbose=0, tol=1e-5)
    # project_soc with a general affine image, and project_multilinear row by row
    As, bs = rng.standard_normal((3, 3)), rng.standard_normal(3)
    zs_ = rng.standard_normal((30, 3)) * 2
    out["gsoc_A"], out["gsoc_b"], out["gsoc_in"] = As, bs, zs_
    out["gsoc_out"] = refproj.project_soc(zs_.copy(), As, bs, rho=1.0, max_iter=100, tol=1e-5)
    Mm, lm, um = rng.standard_normal((2, 3)), np.array([-0.5, -1.0]), np.array([0.7, 0.2])
    xm = rng.standard_normal((25, 3)) * 2
    out["mlin_M"], out["mlin_l"], out["mlin_u"], out["mlin_in"] = Mm, lm, um, xm
    out["mlin_out"] = np.stack([refproj.project_multilinear(r.copy(), Mm, lm, um) for r in xm])
    save("g6_projections.npz", **out)


# ---------------------------------------------------------------------------------------------
# G7: config 5 -- SLS.ADMM_SLS with SOC chance constraints on the controls (unmodified SLS)
# ---------------------------------------------------------------------------------------------
def gen_sls():
    """`notebooks/Double integrator/LQR and SLS with control bounds.ipynb` cells 3-16 at N=50 (SURVEY config 5), for the
    1-D and the 3-D double integrator and a handful of problems that differ in target, control bound and x0 variance.
    Every problem is solved by its own unmodified reference `SLS` object."""
    import contextlib
    import io
    from scipy.stats import norm
    from isls import SLS as RefSLS
    from isls.utils import get_double_integrator_AB as ref_di
    rng = np.random.default_rng(11)
    for tag, nb_dim, nprob in (("d1", 1, 4), ("d3", 3, 2)):
        N, dt = 50, 1.0 / 50
        n, m, p = 2 * nb_dim, nb_dim, nb_dim
        A, B = ref_di(nb_dim, nb_deriv=2, dt=dt)
        u_std = 1e-2
        out = dict(A=A, B=B, N=np.array(N), u_std=np.array(u_std), rho_u=np.array(1e2), alpha=np.array(1.0), tol=np.array(1e-3),
                   max_iter=np.array(50), inner_rho=np.array(10.0), inner_max_iter=np.array(100), inner_threshold=np.array(1e-3))
        targets, bounds, variances, conf = [], [], [], []
        res = dict(du=[], phi_u=[], logs=[], n_it=[], xd=[], A0=[], A1=[], b0=[], b1=[], du0=[], PHI_U=[], K=[], k=[],
                   mc_x0=[], mc_x=[], mc_u=[], fp32_sens=[])
        for b in range(nprob):
            target = np.concatenate([rng.uniform(0.5, 1.5, nb_dim), np.zeros(nb_dim)]) if b else np.concatenate([np.ones(nb_dim), np.zeros(nb_dim)])
            upper_u = [5.0, 4.0, 6.0, 5.5][b % 4]
            var_x0 = [0.01, 0.005, 0.02, 0.01][b % 4]
            psi_inv = norm.ppf([0.95, 0.9, 0.95, 0.82][b % 4])
            sls = RefSLS(n, m, N)
            sls.AB = [A, B]
            zs = np.stack([np.zeros(n), target])
            Qs = np.stack([np.zeros((n, n)), 1e6 * np.eye(n)])
            seq = np.zeros(N, dtype=np.int32)
            seq[N - 1] = 1
            sls.set_quadratic_cost(zs, Qs, seq, u_std)
            # cell 15: SOC rows  psi^-1 ||Sigma^1/2 y|| <= u_max - mu'y  and the mirrored lower bound
            mu = np.zeros(p + 1)
            mu[0] = 1.0
            sigma = np.zeros(p + 1)
            sigma[1:] = var_x0
            Au = np.diag(np.sqrt(sigma))
            A_ = [np.concatenate([Au, (-mu / psi_inv)[None]], 0), np.concatenate([Au, (mu / psi_inv)[None]], 0)]
            b_ = [np.append(np.zeros(p + 1), upper_u / psi_inv), np.append(np.zeros(p + 1), upper_u / psi_inv)]
            project_u = lambda y: refproj.project_set_convex(y, A_, b_, projections=[refproj.project_soc_unit] * 2,   # noqa: E731
                                                             rho=1e1, max_iter=100, threshold=1e-3)
            with contextlib.redirect_stdout(io.StringIO()):
                PHI_U0, du0 = sls.solve_sls()
                sls.l_side_invs = None
                du, phi_u, logs = sls.ADMM_SLS(project_u=project_u, max_iter=50, rho_u=1e2, alpha=1.0, tol=1e-3, verbose=0, log=True)
                K, k = sls.controller(phi_u, du)
            lg = np.full((50, 2), np.nan)
            lg[:len(logs)] = np.stack(logs)
            # fp32 conditioning of THIS problem, measured on the reference itself (like o2_sens of the arm): its own
            # intermediate inverses, its target and its constraint rows are perturbed by random relative errors of fp32
            # rounding size (2^-24) -- what storing the x-step operands in fp32 does at the very least -- and the same
            # ADMM_SLS call is repeated for the same number of iterations; how far du, phi_u move is a floor for any fp32
            # evaluation of this iteration.
            sens_b = np.zeros(2)
            prng = np.random.default_rng(1000 + b)
            jit = lambda a: a * (1.0 + 2.0 ** -24 * prng.uniform(-1.0, 1.0, np.shape(a)))   # noqa: E731
            for _ in range(3):
                sp = RefSLS(n, m, N)
                sp.AB = [A, B]
                sp.set_quadratic_cost(np.stack([np.zeros(n), jit(target)]), Qs, seq, u_std)
                orig_inv = sp.compute_inverses
                sp.compute_inverses = lambda M, f=orig_inv: [jit(np.array(X)) for X in f(M)]
                Ap, bp = [jit(a_) for a_ in A_], [jit(b__) for b__ in b_]
                proj_p = lambda y: refproj.project_set_convex(y, Ap, bp, projections=[refproj.project_soc_unit] * 2,   # noqa: E731
                                                              rho=1e1, max_iter=100, threshold=1e-3)
                with contextlib.redirect_stdout(io.StringIO()):
                    sp.solve_sls()
                    du_p, phi_p, logs_p = sp.ADMM_SLS(project_u=proj_p, max_iter=len(logs), rho_u=1e2, alpha=1.0, tol=1e-3,
                                                      verbose=0, log=True)
                du_k, phi_k = du, phi_u
                if len(logs_p) != len(logs):
                    # the stop iteration is decided by the relative change of a residual at rounding level, i.e. by noise
                    # (SURVEY 8c): compare at the perturbed run's iteration count, which the unperturbed reference reaches too
                    with contextlib.redirect_stdout(io.StringIO()):
                        du_k, phi_k, logs_k = sls.ADMM_SLS(project_u=project_u, max_iter=len(logs_p), rho_u=1e2, alpha=1.0,
                                                           tol=1e-3, verbose=0, log=True)
                    assert len(logs_k) == len(logs_p)
                sens_b[0] = max(sens_b[0], np.max(np.abs(du_p - du_k)) / max(1.0, np.max(np.abs(du_k))))
                sens_b[1] = max(sens_b[1], np.max(np.abs(phi_p[:, :p] - phi_k[:, :p])) / max(1.0, np.max(np.abs(phi_k[:, :p]))))
            res["fp32_sens"].append(sens_b)
            print(tag, "problem", b, "fp32 sensitivity of the reference (du, phi_u)", sens_b)
            x0s = np.zeros((16, n))
            x0s[:, :p] = rng.normal(scale=np.sqrt(var_x0), size=(16, p))
            xm, um = sls.get_trajectory_sls(x0s, K, k, noise_scale=0.0)
            res["du"].append(du), res["phi_u"].append(phi_u), res["logs"].append(lg), res["n_it"].append(len(logs))
            res["xd"].append(np.asarray(sls.xd).reshape(-1)), res["A0"].append(A_[0]), res["A1"].append(A_[1])
            res["b0"].append(b_[0]), res["b1"].append(b_[1]), res["du0"].append(du0), res["PHI_U"].append(PHI_U0)
            res["K"].append(K), res["k"].append(k), res["mc_x0"].append(x0s), res["mc_x"].append(xm), res["mc_u"].append(um)
            targets.append(target), bounds.append(upper_u), variances.append(var_x0), conf.append(psi_inv)
            print(tag, "problem", b, "ADMM_SLS iterations", len(logs), "max |du|", np.max(np.abs(du)))
        out.update({k_: np.stack(v) for k_, v in res.items()})
        out.update(targets=np.stack(targets), upper_u=np.array(bounds), var_x0=np.array(variances), psi_inv=np.array(conf))
        if tag == "d1":   # pin: the notebook's own run (N=100 there; here the N=50 variant of SURVEY config 5)
            out["Sw"], out["Su"] = sls.Sw, sls.Su
        save(f"g7_sls_{tag}.npz", **out)


# ---------------------------------------------------------------------------------------------
# G8: Tassa car-parking problem of notebooks/Tutorial.ipynb (non-quadratic cost through get_Cs)
# ---------------------------------------------------------------------------------------------
def tassa_callbacks(N, dt, dist=2.0):
    """Car-parking problem of Tutorial.ipynb (cells 8, 14, 16) as plain numpy callbacks for the reference solver: dynamics,
    cost, and -- the notebook differentiates with autograd, which is not installed here -- hand-written Jacobians and cost
    derivatives (checked against central finite differences in gen_tassa).
      state [px, py, heading, speed], control [steer, accel];  roll = dt*speed,
      back = roll*cos(steer) + dist - sqrt(dist^2 - (roll*sin(steer))^2),  heading += asin(roll*sin(steer)/dist)
      cost_t = wu.u^2 + wx.H(p_xy, sx) (+ wf.H(x, sf) at the last step),  H(x,s) = sqrt(x^2 + s^2) - s"""
    wu = 1e-2 * np.array([1.0, 0.01])
    wx, sx = 1e-3 * np.ones(2), 0.1 * np.ones(2)
    wf, sf = np.array([0.1, 0.1, 1.0, 0.3]), np.array([0.01, 0.01, 0.01, 1.0])
    last = np.zeros(N)
    last[-1] = 1.0

    def huber(x, s):
        return np.sqrt(x ** 2 + s ** 2) - s

    def forward_model(x, u):
        roll = dt * x[..., 3]
        lat = np.sin(u[..., 0]) * roll
        back = roll * np.cos(u[..., 0]) + dist - np.sqrt(dist ** 2 - lat ** 2)
        return np.stack([x[..., 0] + back * np.cos(x[..., 2]), x[..., 1] + back * np.sin(x[..., 2]),
                         x[..., 2] + np.arcsin(lat / dist), x[..., 3] + u[..., 1] * dt], axis=-1)

    def cost_vec(x, u):                                          # per-step cost of one trajectory [N]
        terminal = last * np.sum(wf * huber(x[-1], sf))
        return terminal + np.sum(wu * u ** 2, axis=-1) + np.sum(wx * huber(x[:, :2], sx), axis=-1)

    def cost(x, u):                                              # [L,N,.] candidates -> [L] (NaN -> 1e6), or one trajectory
        if x.ndim == 2:
            return np.sum(cost_vec(x, u))
        c = np.array([np.sum(cost_vec(xi, ui)) for xi, ui in zip(x, u)])
        c[np.isnan(c)] = 1e6
        return c

    def get_AB(x, u):
        roll, sw, cw = dt * x[:, 3], np.sin(u[:, 0]), np.cos(u[:, 0])
        root = np.sqrt(dist ** 2 - (sw * roll) ** 2)
        back = roll * cw + dist - root
        d_roll, d_steer = cw + sw ** 2 * roll / root, -roll * sw + sw * cw * roll ** 2 / root
        sh, ch = np.sin(x[:, 2]), np.cos(x[:, 2])
        A, B = np.tile(np.eye(4), (x.shape[0], 1, 1)), np.zeros((x.shape[0], 4, 2))
        A[:, 0, 2], A[:, 1, 2] = -back * sh, back * ch
        A[:, 0, 3], A[:, 1, 3], A[:, 2, 3] = d_roll * dt * ch, d_roll * dt * sh, sw / root * dt
        B[:, 0, 0], B[:, 1, 0], B[:, 2, 0], B[:, 3, 1] = d_steer * ch, d_steer * sh, cw * roll / root, dt
        return A, B

    def get_Cs(x, u):                                            # gradient [N,6] and Hessian [N,6,6] per step
        g, H = np.zeros((x.shape[0], 6)), np.zeros((x.shape[0], 6, 6))
        r1 = np.sqrt(x[:, :2] ** 2 + sx ** 2)
        g[:, :2] = wx * x[:, :2] / r1
        H[:, [0, 1], [0, 1]] = wx * sx ** 2 / r1 ** 3
        r2 = np.sqrt(x[-1] ** 2 + sf ** 2)
        g[-1, :4] += wf * x[-1] / r2
        H[-1, np.arange(4), np.arange(4)] += wf * sf ** 2 / r2 ** 3
        g[:, 4:] = 2 * wu * u
        H[:, [4, 5], [4, 5]] = 2 * wu
        return g, H
    par = dict(cu=wu, cx=np.array([wx[0], wx[1], 0, 0]), px=np.array([sx[0], sx[1], 1, 1]), cf=wf, pf=sf)
    return forward_model, cost, cost_vec, get_AB, get_Cs, par


def gen_tassa():
    import contextlib
    import io
    N, dt = 100, 0.03
    f, cost, cost_vec, get_AB, get_Cs, par = tassa_callbacks(N, dt)
    rng = np.random.default_rng(5)
    # hand-written derivatives against central finite differences (the notebook's autograd is not available)
    xt, ut = rng.standard_normal((N, 4)) * np.array([1, 1, 1, 3.0]), rng.standard_normal((N, 2)) * np.array([0.4, 1.0])
    A, B = get_AB(xt, ut)
    cs_, Cs_ = get_Cs(xt, ut)
    h = 1e-6
    for j in range(4):
        e = np.zeros(4); e[j] = h
        assert np.max(np.abs((f(xt + e, ut) - f(xt - e, ut)) / (2 * h) - A[:, :, j])) < 1e-7
        assert np.max(np.abs((cost_vec(xt + e, ut) - cost_vec(xt - e, ut)) / (2 * h) - cs_[:, j])) < 1e-7
        g1, _ = get_Cs(xt + e, ut); g0, _ = get_Cs(xt - e, ut)
        assert np.max(np.abs((g1 - g0)[:, :] / (2 * h) - Cs_[:, :, j])) < 1e-6
    for j in range(2):
        e = np.zeros(2); e[j] = h
        assert np.max(np.abs((f(xt, ut + e) - f(xt, ut - e)) / (2 * h) - B[:, :, j])) < 1e-7
        assert np.max(np.abs((cost_vec(xt, ut + e) - cost_vec(xt, ut - e)) / (2 * h) - cs_[:, 4 + j])) < 1e-7
    out = dict(N=np.array(N), dt=np.array(dt), dist=np.array(2.0), fd_x=xt, fd_u=ut, fd_A=A, fd_B=B, fd_cs=cs_, fd_Cs=Cs_,
               **{"par_" + k: v for k, v in par.items()})
    nprob = 2
    x0s, u0s, res = [], [], dict(cost0=[], x_nom0=[], K0=[], k0=[], cost_log=[], n_it=[], x_fin=[], u_fin=[])
    traces = []
    for b in range(nprob):
        x0 = np.array([1.0, 1.0, 1.5 * np.pi, 0.0]) + (0.0 if b == 0 else 0.05) * rng.standard_normal(4)
        u0 = rng.standard_normal((N, 2)) * 0.1
        x0s.append(x0), u0s.append(u0)

        def fresh():
            obj = ref.iSLS(x_dim=4, u_dim=2, N=N)
            obj.forward_model = f
            obj.cost_function = cost
            x_nom, u_nom = obj.get_trajectory_batch(x0, u0)
            obj.reset()
            obj.nominal_values = x_nom, u_nom
            return obj
        obj = fresh()
        res["cost0"].append(float(obj.cost)), res["x_nom0"].append(obj.x_nom.copy())
        obj.A, obj.B = get_AB(obj.x_nom, obj.u_nom)
        cts, Cts = get_Cs(obj.x_nom, obj.u_nom)
        K, k = obj.backward_pass_DP(Cts=Cts, cts=cts)
        res["K0"].append(K), res["k0"].append(k)
        with contextlib.redirect_stdout(io.StringIO()):
            obj.solve(get_AB, get_Cs, max_iter=6, max_line_search_iter=40, method='dp', verbose=False)
        cl = np.full(8, np.nan)
        cl[:len(obj.cost_log)] = obj.cost_log
        res["cost_log"].append(cl), res["n_it"].append(len(obj.cost_log)), res["x_fin"].append(obj.x_nom.copy()), res["u_fin"].append(obj.u_nom.copy())
        print("tassa problem", b, "iLQR cost log", obj.cost_log)
        # O2 with the control limits of cells 25-27 (box u1 in [-.5,.5], u2 in [-2,2], rho_u = diag(1e-1, 1e-2), 5 ADMM its)
        obj = fresh()
        obj._gen_get_Cs = get_Cs
        lo, hi = np.tile([-0.5, -2.0], N), np.tile([0.5, 2.0], N)
        tr = []
        o2_ilqr_admm(obj, get_AB, None, lambda u: refproj.project_bound(u, lo, hi), None, np.diag([1e-1, 1e-2]), max_iter=3, L=40, J=5,
                     relax=1.0, tol=0.0, trace=tr)
        traces.append(tr)
    out.update(x0=np.stack(x0s), u0=np.stack(u0s), **{k_: np.stack(v) for k_, v in res.items()})
    pack_trace("o2", traces, 4, 2, N, 5, out)
    save("g8_tassa.npz", **out)


# ---------------------------------------------------------------------------------------------
# G9: iSLS.isls_admm (isls/isls.py:503-712, shimmed) on the 3R arm with the chance constraint on the controls of
# notebooks/3DoF robot/State bounds and robust control bounds.ipynb cells 24-26
# ---------------------------------------------------------------------------------------------
def robust_control_rows(q_dim, upper, lower, var_x0, psi_inv):
    """A_, b_ of the two unit-SOC images of the chance constraint on a row [u_nom + d_u, phi_u]."""
    mu = np.zeros(1 + q_dim)
    mu[0] = 1.0
    root = np.diag(np.sqrt(np.concatenate([[0.0], np.full(q_dim, var_x0)])))
    A_ = [np.concatenate([root, (-mu / psi_inv)[None]], axis=0), np.concatenate([root, (mu / psi_inv)[None]], axis=0)]
    b_ = [np.append(np.zeros(1 + q_dim), upper / psi_inv), np.append(np.zeros(1 + q_dim), -lower / psi_inv)]
    return A_, b_


def gen_isls_admm():
    import contextlib
    import io
    from scipy.stats import norm
    out = {}
    N, q_dim = 40, 3
    cfg = P.config3(batch=2, N=N, seed=3)
    cfg["u0"] = np.zeros_like(cfg["u0"])                       # notebook cell 26 starts from zero controls
    upper, lower, var_x0, psi_inv = 6.0, -6.0, 0.1, float(norm.ppf(0.82))
    A_, b_ = robust_control_rows(q_dim, upper, lower, var_x0, psi_inv)
    out.update(upper=np.array(upper), lower=np.array(lower), var_x0=np.array(var_x0), psi_inv=np.array(psi_inv))
    res = {k_: [] for k_ in ("du", "phi_u", "x_nom", "u_nom", "cost_log", "n_outer", "unc_du", "unc_phi_u", "unc_cost_log",
                             "proj_in", "proj_out", "n_proj", "ctl_k", "ctl_K_probe", "mc_x0", "mc_x", "mc_u")}
    probe = np.random.default_rng(5).standard_normal(N * 9)
    for b in range(2):
        calls_in, calls_out = [], []

        def project_u(u, u_nom):
            y = u.copy()
            y[:, 0] += u_nom.flatten()
            calls_in.append(y.copy())
            y = refproj.project_set_convex(y, A_, b_, projections=[refproj.project_soc_unit] * 2, rho=1e1, max_iter=100,
                                           threshold=1e-4, verbose=0)
            calls_out.append(y.copy())
            y[:, 0] -= u_nom.flatten()
            return y
        # unconstrained call (cell 23): feedback columns of the plain iLQR problem
        obj, get_AB = make_ref_isls(cfg, b)
        with contextlib.redirect_stdout(io.StringIO()):
            du, phi_u = obj.isls_admm(q_dim, get_AB, max_line_search=10, k_max=3, max_admm_iter=1, threshold=1e-4, log=True)
        res["unc_du"].append(du), res["unc_phi_u"].append(phi_u)
        res["unc_cost_log"].append(np.array(obj.cost_log, dtype=float)[:4])
        # robust control bounds (cell 26), 3 outer iterations x <= 10 ADMM iterations
        obj, get_AB = make_ref_isls(cfg, b)
        with contextlib.redirect_stdout(io.StringIO()):
            du, phi_u = obj.isls_admm(q_dim, get_AB, max_line_search=30, k_max=3, project_u=project_u, rho_u=1.0,
                                      max_admm_iter=10, threshold=1e-4, verbose=0, log=True)
        print("isls_admm problem", b, "cost log", obj.cost_log, "projection calls", len(calls_in))
        res["du"].append(du), res["phi_u"].append(phi_u), res["x_nom"].append(obj.x_nom.copy()), res["u_nom"].append(obj.u_nom.copy())
        res["cost_log"].append(np.array(obj.cost_log, dtype=float)[:4]), res["n_outer"].append(len(obj.cost_log) - 1)
        # controller of the notebook's cells 23 / 26 (SLS.controller on the iSLS object's Sw, Su) and its Monte-Carlo closed loop
        PHI_U = np.zeros((3 * N, 9 * N))
        PHI_U[:, :q_dim] = phi_u
        K_sls, k_sls = ref.SLS.controller(obj, PHI_U, du)
        x0s = np.tile(obj.x_nom[0:1], (8, 1))
        x0s[:, :q_dim] += np.sqrt(var_x0) * np.random.default_rng(7 + b).standard_normal((8, q_dim))
        x0s[:, 6:] = P.arm_fk(x0s[:, :q_dim])
        mx, mu = obj.get_trajectory_sls(x0s, K_sls, k_sls)
        res["ctl_k"].append(k_sls), res["ctl_K_probe"].append(K_sls @ probe), res["mc_x0"].append(x0s), res["mc_x"].append(mx), res["mc_u"].append(mu)
        # arguments and results of the projection during the first outer iteration (one call per ADMM iteration)
        res["proj_in"].append(np.stack(calls_in[:10])), res["proj_out"].append(np.stack(calls_out[:10])), res["n_proj"].append(len(calls_in))
    out.update(x0=cfg["x0"], **{k_: np.stack(v) for k_, v in res.items()})
    save("g9_isls_admm.npz", **out)


# ---------------------------------------------------------------------------------------------
# G10: batch-form iLQR of iSLS (backward_pass_batch / iterate_once_batch / solve(method='batch'), isls/isls.py:156-228,
# shimmed) on the 3R arm and the car
# ---------------------------------------------------------------------------------------------
def gen_batch_ilqr():
    import contextlib
    import io
    out = {}
    for name, cfg in (("arm", P.config3(batch=2, N=40, seed=3)), ("car", P.config4(batch=2, N=60, seed=2))):
        res = {k_: [] for k_ in ("du0", "cost_log", "n_it", "x_fin", "u_fin")}
        for b in range(2):
            obj, get_AB = make_ref_isls(cfg, b)
            obj.AB = get_AB(obj.x_nom, obj.u_nom)
            res["du0"].append(obj.backward_pass_batch())
            # HEAD's solve() calls get_Cs unconditionally, so the quadratic-cost problem runs through the loop of
            # isls.py:106-132 written out here (the notebook-era solve_ilqr(dp=False))
            with contextlib.redirect_stdout(io.StringIO()):
                for i in range(6):
                    obj.AB = get_AB(obj.x_nom, obj.u_nom)
                    ok = obj.iterate_once_batch(max_line_search=20)
                    if np.abs(np.diff(obj.cost_log[-2:])) < 1e-5 or not ok:
                        break
            cl = np.full(8, np.nan)
            cl[:len(obj.cost_log)] = obj.cost_log
            res["cost_log"].append(cl), res["n_it"].append(len(obj.cost_log)), res["x_fin"].append(obj.x_nom.copy()), res["u_fin"].append(obj.u_nom.copy())
            print("batch iLQR", name, b, obj.cost_log)
        out.update({f"{name}_{k_}": np.stack(v) for k_, v in res.items()})
    save("g10_batch_ilqr.npz", **out)


# ---------------------------------------------------------------------------------------------
# G11: SLS.ADMM_SLS with state AND control chance constraints ("LQR and SLS with state bounds.ipynb" cells 12-17) at N=40
# ---------------------------------------------------------------------------------------------
def gen_sls_state():
    import contextlib
    import io
    from scipy.stats import norm
    from isls import SLS as RefSLS
    from isls.utils import get_double_integrator_AB as ref_di
    N, n, m, p = 40, 2, 1, 1
    A, B = ref_di(1, nb_deriv=2, dt=1.0 / N)
    out = dict(A=A, B=B)
    res = dict(du=[], phi_u=[], logs=[], n_it=[], rho_x=[])
    for b, (target, upper_u, var_x0, conf) in enumerate(((1.0, 3.0, 0.02, 0.9), (0.8, 2.5, 0.01, 0.95))):
        sls = RefSLS(n, m, N)
        sls.AB = [A, B]
        zs, Qs = np.stack([np.zeros(n), [target, 0.0]]), np.stack([np.zeros((n, n)), 1e6 * np.eye(n)])
        seq = np.zeros(N, dtype=np.int32)
        seq[N - 1] = 1
        sls.set_quadratic_cost(zs, Qs, seq, 1e-2)
        psi_inv = norm.ppf(conf)
        mu, sigma = np.array([1.0, 0.0]), np.array([0.0, var_x0])
        Au = np.diag(np.sqrt(sigma))
        A_ = [np.concatenate([Au, (-mu / psi_inv)[None]], 0), np.concatenate([Au, (mu / psi_inv)[None]], 0)]
        b_u = [np.append(np.zeros(2), upper_u / psi_inv), np.append(np.zeros(2), upper_u / psi_inv)]
        # final position within [target - 0.05, target + 0.05], final velocity pinned to 0 (cell 16), all in the chance sense
        b_pos = [np.append(np.zeros(2), (target + 0.05) / psi_inv), np.append(np.zeros(2), -(target - 0.05) / psi_inv)]
        b_vel = [np.append(np.zeros(2), 0.0), np.append(np.zeros(2), 0.0)]
        kw = dict(projections=[refproj.project_soc_unit] * 2, rho=1e1, max_iter=20, threshold=1e-2)
        project_u = lambda y: refproj.project_set_convex(y, A_, b_u, **kw)        # noqa: E731

        def project_x(x):
            x_ = x.copy()
            x_[-2:-1] = refproj.project_set_convex(x_[-2:-1], A_, b_pos, **kw)
            x_[-1:] = refproj.project_set_convex(x_[-1:], A_, b_vel, **kw)
            return x_
        rho_x = np.zeros((N, n, n))
        rho_x[-1, 0, 0] = rho_x[-1, 1, 1] = 1e3
        with contextlib.redirect_stdout(io.StringIO()):
            du, phi_u, logs = sls.ADMM_SLS(project_u=project_u, project_x=project_x, max_iter=30, rho_x=rho_x, rho_u=1e-3,
                                           alpha=1.0, tol=1e-5, verbose=0, log=True)
        lg = np.full((30, 2), np.nan)
        lg[:len(logs)] = np.stack(logs)
        res["du"].append(du), res["phi_u"].append(phi_u), res["logs"].append(lg), res["n_it"].append(len(logs)), res["rho_x"].append(rho_x)
        print("sls state problem", b, "iterations", len(logs), "final residuals", logs[-1], "max |du|", np.max(np.abs(du)))
    out.update({k_: np.stack(v) for k_, v in res.items()})
    out.update(targets=np.array([1.0, 0.8]), upper_u=np.array([3.0, 2.5]), var_x0=np.array([0.02, 0.01]), conf=np.array([0.9, 0.95]))
    save("g11_sls_state.npz", **out)


# ---------------------------------------------------------------------------------------------
# G12: LQT-ADMM with an arbitrary numpy state projection (project_set_convex + Dykstra over quadratic shells):
# "LQR and SLS with spherical obstacle avoidance.ipynb" cells 4-14 at N=60
# ---------------------------------------------------------------------------------------------
def spherical_obstacle_projection(projmod, x_dim, d, centres, radii):
    """project_state of the notebook (cell 12) built from a projections module (the reference's or ours)."""
    lowers = [0.5 * (1.1 * r) ** 2 for r in radii]
    shells = [lambda x, lo=lo, c=c: projmod.project_quadratic(x - c, lo, 1e2) + c for lo, c in zip(lowers, centres)]
    eyes, zeros = [np.eye(x_dim)] * len(radii), [np.zeros(x_dim)] * len(radii)

    def project_state(x):
        x_ = x.reshape(-1, d).copy()
        x_[:, :x_dim] = projmod.project_set_convex(x_[:, :x_dim], eyes, zeros, shells, max_iter=5, verbose=0, threshold=1e-2)
        x_[:, :x_dim] = projmod.project_set_convex_dykstra(x_[:, :x_dim], shells, max_iter=50, verbose=0, tol=1e-5)
        return x_.flatten()
    return project_state


def gen_obstacles():
    import contextlib
    import io
    from isls import SLS as RefSLS
    from isls.utils import get_double_integrator_AB as ref_di
    N, x_dim = 60, 2
    d = 2 * x_dim
    A, B = ref_di(x_dim, nb_deriv=2, dt=1.0 / N)
    sls = RefSLS(d, x_dim, N)
    sls.AB = [A, B]
    zs, Qs = np.stack([np.zeros(d), [1.0, 1.0, 0.0, 0.0]]), np.stack([np.zeros((d, d)), 1e3 * np.eye(d)])
    seq = np.zeros(N, dtype=np.int32)
    seq[N - 1] = 1
    sls.set_quadratic_cost(zs, Qs, seq, 1e-4)
    centres, radii = [np.array([0.5, 0.5]), np.array([0.5, 0.2])], [0.1, 0.15]
    project_state = spherical_obstacle_projection(refproj, x_dim, d, centres, radii)
    rho_x = np.zeros((N, d, d))
    rho_x[:, :x_dim, :x_dim] = np.eye(x_dim)
    with contextlib.redirect_stdout(io.StringIO()):
        # keeping points OUT of a ball is a non-convex projection: the iteration amplifies rounding differences by ~10x per
        # iteration from iteration ~25 on (two runs of the same algorithm with a different summation order agree to 1e-12 for
        # 25 iterations and to nothing after 45), so the vectors stop where the reference's own trace is still reproducible
        xb, ub, logb = sls.ADMM_LQT_Batch(np.zeros(d), project_x=project_state, max_iter=20, rho_x=rho_x, alpha=1.0, tol=1e-3,
                                          verbose=0, log=True)
        xd, ud, Kd, kd, logd = sls.ADMM_LQT_DP(np.zeros(d), project_x=project_state, max_iter=25, rho_x=rho_x, tol=1e-4,
                                               verbose=False, log=True)
    print("obstacles: batch iterations", len(logb), "dp iterations", len(logd), "min distance to centre 0",
          np.min(np.linalg.norm(xd.reshape(N, d)[:, :2] - centres[0], axis=1)))
    save("g12_obstacles.npz", A=A, B=B, batch_x=xb, batch_u=ub, batch_logs=np.stack(logb), dp_x=xd, dp_u=ud, dp_k=kd,
         dp_logs=np.stack(logd), centres=np.stack(centres), radii=np.array(radii))


# ---------------------------------------------------------------------------------------------
# G13: closed loops with process noise (np.random.normal per step) from the unmodified SLS / iSLS, seeded
# ---------------------------------------------------------------------------------------------
def gen_noise():
    c = P.config1(50)
    sls = ref.SLS(2, 1, 50)
    sls.AB = [c["A"], c["B"]]
    sls.set_quadratic_cost(c["zs"], c["Qs"], c["seq"], c["u_std"])
    K, k = sls.solve_dp()
    x0s = np.random.default_rng(5).normal(scale=0.1, size=(7, 2))
    out = dict(K=K, k=k, x0s=x0s)
    np.random.seed(123)
    out["dp_x"], out["dp_u"] = sls.get_trajectory_dp(x0s, K, k, noise_scale=0.05)
    us = np.random.default_rng(6).normal(size=(50, 1))
    np.random.seed(124)
    out["batch_us"] = us
    out["batch_x"], out["batch_u"] = sls.get_trajectory_batch(x0s, us, noise_scale=0.02)
    # iSLS: the arm through its own numpy forward model, open loop with noise and DP closed loop about zero gains
    cfg = P.config3(batch=1, N=100, seed=0)
    f, _ = P.model_callbacks(cfg)
    isl = RefISLS(9, 3, 100)
    isl.forward_model = f
    x0 = cfg["x0"][0]
    np.random.seed(125)
    out["arm_x0"], out["arm_us"] = x0, cfg["u0"][0]
    out["arm_batch_x"], out["arm_batch_u"] = isl.get_trajectory_batch(x0, cfg["u0"][0], noise_scale=0.01)
    Kz, kz = 0.1 * np.random.default_rng(7).normal(size=(100, 3, 9)), cfg["u0"][0]
    np.random.seed(126)
    out["arm_K"] = Kz
    out["arm_dp_x"], out["arm_dp_u"] = isl.get_trajectory_dp(x0, Kz, kz, noise_scale=0.01)
    # a BATCH of initial states (2-D x0): the reference's iSLS returns only trajectory 0 of the batch (the `x0.ndim == 2` test of
    # isls_base.py:39-42,68-71 is inverted); the noise is still drawn for all of them, so row 0 pins the batched loop
    x0s2 = x0[None] + np.random.default_rng(8).normal(scale=0.05, size=(3, 9))
    np.random.seed(127)
    out["arm_x0s2"] = x0s2
    out["arm_dp2_x0"], out["arm_dp2_u0"] = isl.get_trajectory_dp(x0s2, Kz, kz, noise_scale=0.01)
    save("g13_noise.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["di1d", "di3d", "arm", "car", "proj", "sls", "tassa", "isls_admm", "batch_ilqr", "sls_state", "obstacles", "noise"]
    for w in which:
        {"di1d": gen_di1d, "di3d": gen_di3d, "arm": gen_arm, "car": gen_car, "proj": gen_projections, "sls": gen_sls,
         "tassa": gen_tassa, "isls_admm": gen_isls_admm, "batch_ilqr": gen_batch_ilqr, "sls_state": gen_sls_state, "obstacles": gen_obstacles, "noise": gen_noise}[w]()
