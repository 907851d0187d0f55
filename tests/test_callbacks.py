"""The reference's callback surface: `forward_model(x, u)`, `get_AB(x, u)`, `cost_function(x, u)` and
`get_Cs(x, u)` as plain Python callables (isls/isls.py:95-110,153,332,360), and process noise in `get_trajectory_*`
(isls/isls_base.py:28-71, isls/sls_base.py:61-105).  With callables the line search runs on the host (isls/hostpath.py)
while the Riccati passes and the ADMM update stay on the device; results must equal the reference's golden traces exactly
as the all-device path does, and the device path itself."""
import numpy as np
import pytest

import isls_problems as P
from test_isls_api import TOL, _check_final, _tols, make_isls, rel

pytestmark = pytest.mark.gpu


def _callback_isls(cfg, bsel):
    """iSLS with the NOTEBOOK's own numpy model function (a plain callable, no isls.models descriptor)."""
    import isls
    f, get_AB = P.model_callbacks(cfg)
    s = isls.iSLS(cfg["n"], cfg["m"], cfg["N"], batch=len(bsel))
    s.forward_model = lambda x, u: f(x, u)                       # a lambda: nothing the front end could recognise
    s.set_cost_variables(cfg["zs"], cfg["Qs"], cfg["seq"], cfg["u_std"])
    xs, us = zip(*[P.initial_nominal(cfg, b) for b in bsel])
    s.reset()
    s.nominal_values = (xs[0], us[0]) if len(bsel) == 1 else (np.stack(xs), np.stack(us))
    return s, get_AB


def test_ilqr_admm_with_notebook_callables(golden):
    """Config 3 / config 4 exactly as the notebooks call them -- python f, python get_AB, box projections -- against the
    reference's O2 traces (the same check as test_ilqr_admm_arm_and_car on the all-device path)."""
    from isls import Box
    for name, cfg in (("g4_arm3r.npz", P.config3(batch=2, N=100, seed=0)), ("g5_car.npz", P.config4(batch=2, N=200, seed=0))):
        g = golden(name)
        s, get_AB = _callback_isls(cfg, [0, 1])
        L = cfg.get("max_line_search", 20)
        s.ilqr_admm(get_AB, project_x=Box(cfg["x_lo"], cfg["x_hi"]), project_u=Box(cfg["u_lo"], cfg["u_hi"]), max_iter=3,
                    max_line_search_iter=L, max_admm_iter=cfg["max_admm_iter"], rho_x=cfg["rho_x"], rho_u=cfg["rho_u"],
                    alpha=1.0, tol=0.0)
        _check_final(s, g, "o2", [0, 1], 3, cfg["max_admm_iter"], _tols(g, "o2"))


def test_ilqr_with_notebook_callables_matches_device_path():
    """`solve` (plain iLQR, NaN rule + acceptance test) through the host line search equals the rollout kernel's result."""
    cfg = P.config4(batch=3, N=200, seed=0)
    s, get_AB = _callback_isls(cfg, [0, 1, 2])
    s.solve(get_AB, max_iter=4, max_line_search_iter=20)
    d = make_isls(cfg, [0, 1, 2])
    d.solve(max_iter=4, max_line_search_iter=20)
    assert rel(s.x_nom, d.x_nom) < 1e-9 and rel(s.u_nom, d.u_nom) < 1e-9 and rel(s.cost, d.cost) < 1e-9
    # rollout_DP through the callable (batch of one, the reference's own signature)
    s1, _ = _callback_isls(cfg, [1])
    K, k = s1.K, s1.k
    x_log, u_log = s1.rollout_DP(K, k[None] * np.array([1.0, 0.5])[:, None, None])
    assert x_log.shape == (2, 200, 4) and np.allclose(x_log[:, 0], s1.x_nom[0])


def test_tassa_with_callable_cost_and_get_Cs(golden):
    """Tutorial.ipynb with everything given as callables: model, get_AB, cost_function and get_Cs (wrapped in lambdas so that
    nothing is recognised as a built-in): iLQR and iLQR-ADMM against the reference's run of the notebook (golden G8, the same
    checks as test_tassa_car_parking_api on the all-device path)."""
    from isls import Box, costs, iSLS, models
    g = golden("g8_tassa.npz")
    N = int(g["N"])
    cost = costs.PseudoHuber(g["par_cu"], g["par_cx"], g["par_px"], g["par_cf"], g["par_pf"])
    mdl = models.TassaCar(float(g["dt"]), float(g["dist"]))
    get_AB, get_Cs = (lambda x, u: mdl.get_AB(x, u)), (lambda x, u: cost.get_Cs(x, u))

    def fresh():
        s = iSLS(4, 2, N, batch=2)
        s.forward_model = lambda x, u: mdl(x, u)
        s.cost_function = lambda x, u: cost(x, u)
        s.nominal_values = g["x_nom0"], g["u0"]
        return s
    s = fresh()
    assert rel(s.cost, g["cost0"]) < 1e-12
    s.solve(get_AB, get_Cs=get_Cs, max_iter=6, max_line_search_iter=40, method='dp')
    assert rel(s.cost, g["cost_log"][:, 6]) < 1e-7 and rel(s.x_nom, g["x_fin"]) < 1e-5 and rel(s.u_nom, g["u_fin"]) < 1e-5
    s = fresh()
    s.ilqr_admm(get_AB, get_Cs=get_Cs, project_u=Box(np.array([-0.5, -2.0]), np.array([0.5, 2.0])), max_iter=3,
                max_line_search_iter=40, max_admm_iter=5, rho_u=np.diag([1e-1, 1e-2]), tol=0.0)
    _check_final(s, g, "o2", [0, 1], 3, 5, {k: 1e-7 for k in ("xx", "xu", "K", "cost")})


def test_isls_admm_with_callables_matches_device_path():
    """`isls_admm` with a callable model + get_AB (and get_Cs of the via-point cost written out by hand) equals the all-device
    call on the same arm problem (isls/isls.py:503-505,548-560)."""
    import isls
    from isls import models
    from isls.projections import chance_constraint_rows
    from scipy.stats import norm
    cfg = P.config3(batch=2, N=100, seed=0)
    cfg["u0"] = np.zeros_like(cfg["u0"])
    cs = chance_constraint_rows(3, 6.0, -6.0, 0.1, float(norm.ppf(0.82)), rho=10.0, max_iter=100, threshold=1e-4)
    kw = dict(max_line_search=10, k_max=2, project_u=cs, rho_u=1.0, max_admm_iter=4, threshold=0.0)
    s, get_AB = _callback_isls(cfg, [0, 1])
    zs, Qs, seq, u_std = cfg["zs"], cfg["Qs"], cfg["seq"], cfg["u_std"]

    def get_Cs(x, u):                                           # gradient / Hessian of the via-point cost (no 1/2: 2 Q, 2 R)
        n, m, N = 9, 3, 100
        Cs, cs_ = np.zeros((N, n + m, n + m)), np.zeros((N, n + m))
        Cs[:, :n, :n], Cs[:, n:, n:] = 2 * Qs[seq], 2 * u_std * np.eye(m)
        cs_[:, :n], cs_[:, n:] = np.einsum("tij,tj->ti", 2 * Qs[seq], x - zs[seq]), 2 * u_std * u
        return cs_, Cs
    du_s, phi_s = s.isls_admm(3, get_AB, get_Cs=get_Cs, **kw)
    d = make_isls(cfg, [0, 1])
    du_d, phi_d = d.isls_admm(3, None, **kw)
    assert rel(du_s, du_d) < 1e-7 and rel(phi_s, phi_d) < 1e-7 and rel(s.x_nom, d.x_nom) < 1e-7


def test_process_noise_reproduces_the_seeded_reference(golden):
    """get_trajectory_dp / get_trajectory_batch with noise_scale: numpy's global generator, one draw per step in the
    reference's order, so np.random.seed(s) gives the reference's trajectories (golden G13, produced by the reference)."""
    import isls
    from isls import models
    g = golden("g13_noise.npz")
    c = P.config1(50)
    sls = isls.SLS(2, 1, 50)
    sls.AB = [c["A"], c["B"]]
    sls.set_quadratic_cost(c["zs"], c["Qs"], c["seq"], c["u_std"])
    np.random.seed(123)
    x, u = sls.get_trajectory_dp(g["x0s"], g["K"], g["k"], noise_scale=0.05)
    assert rel(x, g["dp_x"]) < 1e-12 and rel(u, g["dp_u"]) < 1e-12
    np.random.seed(124)
    x, u = sls.get_trajectory_batch(g["x0s"], g["batch_us"], noise_scale=0.02)
    assert rel(x, g["batch_x"]) < 1e-12 and rel(u, g["batch_u"]) < 1e-12
    arm = isls.iSLS(9, 3, 100)
    arm.forward_model = models.Planar3R(0.01)
    np.random.seed(125)
    x, u = arm.get_trajectory_batch(g["arm_x0"], g["arm_us"], noise_scale=0.01)
    assert rel(x, g["arm_batch_x"]) < 1e-12 and rel(u, g["arm_batch_u"]) < 1e-12
    np.random.seed(126)
    x, u = arm.get_trajectory_dp(g["arm_x0"], g["arm_K"], g["arm_us"], noise_scale=0.01)
    assert rel(x, g["arm_dp_x"]) < 1e-12 and rel(u, g["arm_dp_u"]) < 1e-12
    # a batch of initial states: the reference's iSLS keeps trajectory 0 only (inverted `x0.ndim` test, isls_base.py:39-42,
    # 68-71), this package returns the whole batch (documented deviation); row 0 is the reference's
    np.random.seed(127)
    x, u = arm.get_trajectory_dp(g["arm_x0s2"], g["arm_K"], g["arm_us"], noise_scale=0.01)
    assert x.shape == (3, 100, 9) and u.shape == (3, 100, 3)
    assert rel(x[0], g["arm_dp2_x0"]) < 1e-12 and rel(u[0], g["arm_dp2_u0"]) < 1e-12
