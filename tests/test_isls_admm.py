"""iSLS.isls_admm (isls/isls.py:503-712): feedback columns [d, phi] with ADMM on their rows.

CPU: the dense numpy restatement (oracle/isls_admm_dense.py) against the reference's own outputs (g9, 3R arm, chance
constraint on the controls of the robust-control notebook).  GPU: the DP-form HIP path (isls_riccati_gain/ff,
isls_columns_rollout, isls_rollout_ls, isls_project_rows, isls_columns_admm) against that oracle and against g9.

Tolerances.  The reference solves normal equations (Su'Q Su + R) whose condition number is ~1e10 for these costs
(Q = 1e6, R = 1e-4), so the reference's own columns carry a relative error of ~1e-6 (two runs of the same dense algebra with a
different summation order differ by that much, see UNC_TOL); with the ADMM weight Rr = I added the system is well conditioned
and everything agrees to 1e-8 relative after 3 outer x 10 ADMM iterations.  fp32 is checked on the kernels only (1e-4)."""
import numpy as np
import pytest

import isls_problems as P
from helpers import problem_arrays

CON_TOL = 1e-8       # constrained problem (well conditioned), relative to the largest entry
UNC_TOL = 2e-5       # unconstrained columns: limited by the conditioning of the reference's dense solve


def arm_cfg(batch=2, N=40):
    cfg = P.config3(batch=batch, N=N, seed=3)
    cfg["u0"] = np.zeros_like(cfg["u0"])
    return cfg


def control_sets(g):
    from isls.projections import chance_constraint_rows
    return chance_constraint_rows(3, float(g["upper"]), float(g["lower"]), float(g["var_x0"]), float(g["psi_inv"]),
                                  rho=10.0, max_iter=100, threshold=1e-4)


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, float(np.max(np.abs(b)))))


# ---------------------------------------------------------------------------------------------------------
# CPU: oracle pinned by the reference's outputs
# ---------------------------------------------------------------------------------------------------------
def test_dense_oracle_matches_reference(oracle, golden):
    from oracle.isls_admm_dense import DenseIslsAdmm, shifted_sets_projection
    g = golden("g9_isls_admm.npz")
    cfg, cs = arm_cfg(), control_sets(g)
    for b in range(2):
        pa = problem_arrays(cfg, [b])
        d = DenseIslsAdmm(oracle, pa, 3)
        du, phi = d.solve(3, 1, 10)
        assert rel(du, g["unc_du"][b]) < UNC_TOL and rel(phi, g["unc_phi_u"][b]) < UNC_TOL
        assert rel(d.cost_log, g["unc_cost_log"][b]) < 1e-9
        calls = []
        proj = shifted_sets_projection(oracle, cs)

        def project_u(rows, nominal):
            out = proj(rows, nominal)
            calls.append((rows + np.pad(nominal.reshape(-1, 1), ((0, 0), (0, 3))), out + np.pad(nominal.reshape(-1, 1), ((0, 0), (0, 3)))))
            return out
        d = DenseIslsAdmm(oracle, pa, 3, project_u=project_u, rho_u=1.0, threshold=1e-4)
        du, phi = d.solve(3, 10, 30)
        assert len(calls) == int(g["n_proj"][b])                              # same number of ADMM iterations
        for i in range(10):                                                   # first outer iteration, call by call
            assert rel(calls[i][0], g["proj_in"][b][i]) < 1e-9 and rel(calls[i][1], g["proj_out"][b][i]) < 1e-9
        assert rel(du, g["du"][b]) < 1e-9 and rel(phi, g["phi_u"][b]) < 1e-9
        assert rel(d.cost_log, g["cost_log"][b]) < 1e-10
        assert rel(d.x_nom, g["x_nom"][b]) < 1e-10 and rel(d.u_nom, g["u_nom"][b]) < 1e-10
        # controller + Monte-Carlo closed loop of the notebook (cells 23 / 26).  The reference returns only the FIRST of the
        # M trajectories for a batch of initial states (inverted `x0.ndim` test, isls_base.py:39-42); g9 holds that one.
        from oracle.isls_admm_dense import controller
        K, k = controller(d.Sw, d.Su, phi, du, 9)
        probe = np.random.default_rng(5).standard_normal(40 * 9)
        assert rel(K @ probe, g["ctl_K_probe"][b]) < 1e-8 and rel(k, g["ctl_k"][b]) < 1e-8
        x0 = np.ascontiguousarray(g["mc_x0"][b])
        xl, ul = np.zeros((8, 40, 9)), np.zeros((8, 40, 3))
        oracle.dense_closed_loop(pa["model"], pa["model_par"], np.ascontiguousarray(K), k, x0, xl, ul, xhat=d.x_nom.copy(), uhat=d.u_nom.copy())
        assert rel(xl[0], g["mc_x"][b]) < 1e-8 and rel(ul[0], g["mc_u"][b]) < 1e-8


def test_transfer_matrices_reproduce_rollout(oracle):
    """x = Sw[:, :n] x0 + Su u of the dense restatement equals stepping the linearised dynamics."""
    from oracle.isls_admm_dense import transfer_matrices
    rng = np.random.default_rng(0)
    N, n, m = 7, 4, 2
    A, B = rng.standard_normal((N, n, n)), rng.standard_normal((N, n, m))
    Sw, Su = transfer_matrices(A, B)
    x0, u = rng.standard_normal(n), rng.standard_normal((N, m))
    x = [x0]
    for t in range(N - 1):
        x.append(A[t] @ x[-1] + B[t] @ u[t])
    assert np.allclose(Sw[:, :n] @ x0 + Su @ u.reshape(-1), np.concatenate(x), rtol=1e-12, atol=1e-12)
    assert not Su[:, -m:].any()                                               # the last control never acts (SURVEY 8a quirk i)


# ---------------------------------------------------------------------------------------------------------
# GPU: kernels
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-10), (np.float32, 1e-4)])
@pytest.mark.parametrize("n,m,C", [(6, 3, 4), (6, 3, 2), (4, 2, 3), (9, 3, 4), (9, 3, 2), (2, 1, 3), (2, 1, 2)])
def test_columns_rollout_kernel(n, m, C, dtype, tol):
    _check_columns_rollout(n, m, C, 12, dtype, tol)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [2, 3, 5, 9])
def test_columns_rollout_short_horizons(N):
    """horizons shorter than / not a multiple of the prefetch ring (padded dead steps, clamped fetches)"""
    _check_columns_rollout(6, 3, 4, N, np.float64, 1e-10)


def _check_columns_rollout(n, m, C, N, dtype, tol):
    import torch
    from isls.engine import kernels
    from oracle.isls_admm_dense import transfer_matrices
    rng = np.random.default_rng(n * 10 + m)
    B = 5
    A = (np.eye(n) + 0.1 * rng.standard_normal((B, N, n, n))).astype(dtype)
    Bm = (0.3 * rng.standard_normal((B, N, n, m))).astype(dtype)
    K = (0.2 * rng.standard_normal((B, N, m, n))).astype(dtype)
    k = rng.standard_normal((C, B, N, m)).astype(dtype)
    Rr = (np.eye(m) * 0.7 + 0.05 * np.ones((m, m))).astype(dtype)[None]
    Cuu = (2 * 0.3 * np.eye(m) + 2 * Rr).astype(dtype)                         # [1,m,m]: shared over batch and time
    c0u = rng.standard_normal((B, N, m)).astype(dtype)
    zu, lu = rng.standard_normal((C, B, N, m)).astype(dtype), rng.standard_normal((C, B, N, m)).astype(dtype)
    active = np.ones(B, dtype=np.int32)
    active[3] = 0
    dev = lambda a: torch.as_tensor(a, device="cuda")                          # noqa: E731
    dx, du = torch.full((C, B, N, n), 7.0, dtype=dev(A).dtype, device="cuda"), torch.full((C, B, N, m), 7.0, dtype=dev(A).dtype, device="cuda")
    kernels().columns_rollout(dev(A), dev(Bm), dev(Cuu), dev(c0u), dev(K), dev(k), dx, du, Rr=dev(Rr), zu=dev(zu), lu=dev(lu),
                              active=dev(active))
    torch.cuda.synchronize()
    dx, du = dx.cpu().numpy(), du.cpu().numpy()
    assert (dx[:, 3] == 7).all() and (du[:, 3] == 7).all()                     # inactive problem untouched
    for b in (0, 1, 2, 4):
        Sw, Su = transfer_matrices(A[b].astype(np.float64), Bm[b].astype(np.float64))
        for c in range(C):
            x0 = np.zeros(n)
            if c:
                x0[c - 1] = 1.0
            # closed loop on the linear model, then the dense identity dx = Su du + Sx (isls.py:589-590)
            x, us = x0.copy(), []
            for t in range(N - 1):
                us.append(K[b, t].astype(np.float64) @ x + k[c, b, t])
                x = A[b, t].astype(np.float64) @ x + Bm[b, t].astype(np.float64) @ us[-1]
            g = (c == 0) * c0u[b, N - 1].astype(np.float64) - 2 * Rr[0].astype(np.float64) @ (zu[c, b, N - 1].astype(np.float64) - lu[c, b, N - 1])
            us.append(-np.linalg.solve(Cuu[0].astype(np.float64), g))
            us = np.array(us)
            assert rel(du[c, b], us) < tol
            assert rel(dx[c, b].reshape(-1), Sw[:, :n] @ x0 + Su @ us.reshape(-1)) < tol * 10


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 1e-5)])
def test_columns_admm_kernel(dtype, tol):
    import torch
    from isls.engine import kernels
    rng = np.random.default_rng(5)
    B, N, n, m, C = 3, 9, 4, 2, 3
    relax = 0.8
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a), device="cuda")    # noqa: E731
    mk = lambda *s: rng.standard_normal(s).astype(dtype)                       # noqa: E731
    host, blocks = {}, {}
    for key, d in (("x", n), ("u", m)):
        h = dict(xx=mk(C, B, N, d), z=mk(C, B, N, d), l=mk(C, B, N, d), W=(np.eye(d) * 0.5 + 0.1 * mk(N, d, d)).astype(dtype), nom=mk(B, N, d))
        host[key] = h
        blocks[key] = dict(xx=dev(h["xx"]), z=dev(h["z"]), l=dev(h["l"]), z_prev=torch.zeros(C, B, N, d, dtype=tdt, device="cuda"),
                           work=torch.zeros(B, N * d, C, dtype=tdt, device="cuda"), W=dev(h["W"]), nom=dev(h["nom"]))
    res, res_prev = torch.zeros(B, 2, dtype=tdt, device="cuda"), torch.full((B, 2), 1e6, dtype=tdt, device="cuda")
    active, iters = torch.ones(B, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")
    kern, dims = kernels(), (B, N, n, m, C)
    kern.columns_admm(0, dims, res, res_prev, x=blocks["x"], u=blocks["u"], relax=relax, active=active)
    prim = dual = 0.0
    for key, d in (("x", n), ("u", m)):
        h = host[key]
        arg = relax * h["xx"].astype(np.float64) + (1 - relax) * h["z"] + h["l"]                  # [C,B,N,d]
        arg[0] += h["nom"]
        rows = np.transpose(arg, (1, 2, 3, 0)).reshape(B, N * d, C)
        assert rel(blocks[key]["work"].cpu().numpy(), rows) < tol
        z_new = np.tanh(rows)                                                  # any row map stands in for the projection
        blocks[key]["work"].copy_(dev(z_new.astype(dtype)))
        zc = np.transpose(z_new.reshape(B, N, d, C), (3, 0, 1, 2)).copy()
        zc[0] -= h["nom"]
        r = h["xx"] - zc
        h["exp"] = (zc, h["l"] + r)
        W = h["W"].astype(np.float64)
        prim = prim + (np.einsum("tij,cbtj->cbti", W, r) ** 2).sum(axis=(0, 2, 3)) ** 0.5
        dual = dual + (np.einsum("tij,cbtj->cbti", W, zc - h["z"]) ** 2).sum(axis=(0, 2, 3)) ** 0.5
    kern.columns_admm(1, dims, res, res_prev, x=blocks["x"], u=blocks["u"], relax=relax, tol_abs=1e-3, tol_rel=1e-3, active=active,
                      iters=iters)
    torch.cuda.synchronize()
    for key in ("x", "u"):
        assert rel(blocks[key]["z"].cpu().numpy(), host[key]["exp"][0]) < tol * 10
        assert rel(blocks[key]["l"].cpu().numpy(), host[key]["exp"][1]) < tol * 10
    out = res.cpu().numpy()
    assert rel(out[:, 0], prim) < tol * 100 and rel(out[:, 1], dual) < tol * 100
    assert (iters.cpu().numpy() == 1).all() and (active.cpu().numpy() == 1).all()
    assert np.array_equal(res_prev.cpu().numpy(), out)


# ---------------------------------------------------------------------------------------------------------
# GPU: the solver against the dense oracle and the reference's outputs
# ---------------------------------------------------------------------------------------------------------
def make_arm(cfg, bsel):
    from test_isls_api import make_isls
    return make_isls(cfg, bsel)


@pytest.mark.gpu
def test_isls_admm_unconstrained_columns(oracle, golden):
    g = golden("g9_isls_admm.npz")
    cfg = arm_cfg()
    s = make_arm(cfg, [0, 1])
    du, phi = s.isls_admm(3, None, max_line_search=10, k_max=3, max_admm_iter=1, threshold=1e-4)
    assert du.shape == (2, 120) and phi.shape == (2, 120, 3)
    for b in range(2):
        assert rel(du[b], g["unc_du"][b]) < UNC_TOL and rel(phi[b], g["unc_phi_u"][b]) < UNC_TOL
        assert rel(np.array(s.cost_log)[:, b], g["unc_cost_log"][b]) < 1e-7
    assert (s.admm_iters == 1).all()


@pytest.mark.gpu
def test_isls_admm_robust_control_bounds(oracle, golden):
    from oracle.isls_admm_dense import DenseIslsAdmm, shifted_sets_projection
    g = golden("g9_isls_admm.npz")
    cfg, cs = arm_cfg(), control_sets(g)
    s = make_arm(cfg, [0, 1])
    du, phi = s.isls_admm(3, None, max_line_search=30, k_max=3, project_u=cs, rho_u=1.0, max_admm_iter=10, threshold=1e-4)
    for b in range(2):
        assert rel(du[b], g["du"][b]) < CON_TOL and rel(phi[b], g["phi_u"][b]) < CON_TOL
        assert rel(np.array(s.cost_log)[:, b], g["cost_log"][b]) < CON_TOL
        assert rel(s.x_nom[b], g["x_nom"][b]) < CON_TOL and rel(s.u_nom[b], g["u_nom"][b]) < CON_TOL
        d = DenseIslsAdmm(oracle, problem_arrays(cfg, [b]), 3, project_u=shifted_sets_projection(oracle, cs), rho_u=1.0, threshold=1e-4)
        d.solve(3, 10, 30)
        assert rel(s._dx_columns[b], d.x_x) < CON_TOL                         # [d_x, phi_x] of the last x-step
        assert len(d.logs[-1]) == s.admm_iters[b]
        assert rel(s.admm_logs[:s.admm_iters[b], b], np.array(d.logs[-1])) < 1e-6
    assert (s.outer_iters == g["n_outer"]).all()
    # controller + Monte-Carlo closed loop (notebook cells 23 / 26) against the reference's outputs and the oracle
    PHI_U = np.zeros((2, 120, 360))
    PHI_U[:, :, :3] = phi
    K, k = s.controller(PHI_U, du)
    probe = np.random.default_rng(5).standard_normal(40 * 9)
    pa = problem_arrays(cfg, [0])
    for b in range(2):
        assert rel(K[b] @ probe, g["ctl_K_probe"][b]) < 1e-7 and rel(k[b], g["ctl_k"][b]) < 1e-7
        x, u = s.get_trajectory_sls(g["mc_x0"][b], K[b], k[b], problem=b)
        assert x.shape == (8, 40, 9) and u.shape == (8, 40, 3)
        assert rel(x[0], g["mc_x"][b]) < 1e-7 and rel(u[0], g["mc_u"][b]) < 1e-7
        xl, ul = np.zeros((8, 40, 9)), np.zeros((8, 40, 3))
        oracle.dense_closed_loop(pa["model"], pa["model_par"], np.ascontiguousarray(K[b]), np.ascontiguousarray(k[b]),
                                 np.ascontiguousarray(g["mc_x0"][b]), xl, ul, xhat=s.x_nom[b].copy(), uhat=s.u_nom[b].copy())
        assert rel(x, xl) < 1e-10 and rel(u, ul) < 1e-10


@pytest.mark.gpu
def test_isls_admm_host_get_AB_with_recorded_iteration(golden):
    """The reference's calling convention -- `get_AB` a host callable, the forward model a device descriptor, the projection a
    device set -- keeps the recorded ADMM iteration (HIP graph) in use across outer iterations: the linearisation a callback
    returns per outer iteration must land in the buffers the recording points at.  Same results as the built-in linearisation
    (`get_AB=None`) up to the rounding of the host / device Jacobians (CON_TOL; stale addresses give O(1) differences)."""
    g = golden("g9_isls_admm.npz")
    cfg, cs = arm_cfg(), control_sets(g)
    kw = dict(max_line_search=30, k_max=3, project_u=cs, rho_u=1.0, max_admm_iter=4, threshold=0.0)
    s0 = make_arm(cfg, [0, 1])
    du0, phi0 = s0.isls_admm(3, None, **kw)
    s1 = make_arm(cfg, [0, 1])
    mdl = s1.forward_model
    calls = []

    def get_AB(x, u):
        calls.append(1)
        return mdl.get_AB(x, u)
    du1, phi1 = s1.isls_admm(3, get_AB, **kw)
    assert len(calls) == 2 * 3                                               # one callback per problem and outer iteration
    assert rel(du1, du0) < CON_TOL and rel(phi1, phi0) < CON_TOL
    assert rel(s1.x_nom, s0.x_nom) < CON_TOL and rel(s1.u_nom, s0.u_nom) < CON_TOL
    assert rel(np.array(s1.cost_log), np.array(s0.cost_log)) < CON_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("nseg", ["1", "4"])
def test_feedback_columns_in_one_launch_equal_one_launch_per_column(golden, monkeypatch, nseg):
    """isls_ff_args ncol: the C = 1 + dim feed-forward passes of an ADMM iteration as ONE launch on the shared records
    (blockIdx.z = column; sequential and time-parallel form) against one launch per column: the same kernel on the same
    operands, so every output is bit-identical."""
    g = golden("g9_isls_admm.npz")
    cfg, cs = arm_cfg(), control_sets(g)
    kw = dict(max_line_search=30, k_max=2, project_u=cs, rho_u=1.0, max_admm_iter=4, threshold=0.0)
    monkeypatch.setenv("ISLS_FF_NSEG", nseg)
    out = {}
    # ... and the C driver of the ADMM iteration (isls_columns_iteration_*: one call per iteration) against the same launches
    # made one by one from the host
    for mode, drv in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("ISLS_ADMM_FF_COLUMNS", mode)
        monkeypatch.setenv("ISLS_ADMM_DRIVER", drv)
        s = make_arm(cfg, [0, 1])
        du, phi = s.isls_admm(3, None, **kw)
        out[mode + drv] = (du, phi, np.array(s.x_nom), np.array(s.u_nom), np.array(s.cost_log), s.admm_logs)
    for key in ("01", "10", "00"):
        for a, b in zip(out["11"], out[key]):
            assert np.array_equal(a, b), key


@pytest.mark.gpu
def test_isls_admm_callable_projection_equals_device_sets(oracle, golden):
    """project_u given as the notebook's numpy closure (host round trip) runs the same iteration as the ConvexSets route."""
    g = golden("g9_isls_admm.npz")
    cfg, cs = arm_cfg(), control_sets(g)
    s = make_arm(cfg, [0, 1])
    du_d, phi_d = s.isls_admm(3, None, max_line_search=30, k_max=2, project_u=cs, rho_u=1.0, max_admm_iter=6, threshold=1e-4)

    def project_u(rows, u_nom):                                                # cell 25 of the robust-control notebook
        y = rows.copy()
        y[:, 0] += u_nom.flatten()
        y = cs(y.reshape(-1)).reshape(rows.shape)
        y[:, 0] -= u_nom.flatten()
        return y
    s2 = make_arm(cfg, [0, 1])
    du_h, phi_h = s2.isls_admm(3, None, max_line_search=30, k_max=2, project_u=project_u, rho_u=1.0, max_admm_iter=6, threshold=1e-4)
    assert rel(du_h, du_d) < 1e-8 and rel(phi_h, phi_d) < 1e-8


@pytest.mark.gpu
def test_isls_admm_single_problem_shapes_and_state_rows(oracle):
    """batch == 1 returns the reference's shapes; project_x (rows [x_nom + d_x, phi_x]) takes the same route."""
    from isls.projections import ConvexSets, SET_BOX
    from oracle.isls_admm_dense import DenseIslsAdmm, shifted_sets_projection
    cfg = P.config2(batch=1, N=30, seed=1)
    from test_isls_api import make_isls
    s = make_isls(cfg, [0])
    C = 3
    par = np.concatenate([np.full(C, -0.4), np.full(C, 0.4)])                   # box on every entry of a row [x + d_x, phi_x]
    cs = ConvexSets(C, (0, C), [dict(kind=SET_BOX, dim=C, A=np.eye(C), b=np.zeros(C), par=par)], rho=5.0, max_iter=50, threshold=1e-6)
    du, phi = s.isls_admm(2, None, max_line_search=10, k_max=2, project_x=cs, rho_x=0.5, max_admm_iter=5, threshold=1e-6)
    assert du.shape == (30 * 3,) and phi.shape == (30 * 3, 2)
    d = DenseIslsAdmm(oracle, problem_arrays(cfg, [0]), 2, project_x=shifted_sets_projection(oracle, cs), rho_x=0.5, threshold=1e-6)
    du_o, phi_o = d.solve(2, 5, 10)
    assert rel(du, du_o) < 1e-7 and rel(phi, phi_o) < 1e-7
    assert rel(s.cost_log[-1], d.cost_log[-1]) < 1e-8
