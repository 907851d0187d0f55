"""BASELINE.json's configs 3, 4 and 5 at their full per-GPU sizes through size-independent properties (the oracle would need
minutes for them): batch invariance (a problem solved inside the big batch equals the same problem solved in a batch of a
few, bit for bit -- slots / workgroups never interact), exact feasibility of the consensus variables, finite decreasing
costs, no status bits, non-growing ADMM residuals.  Config 2 has the same test in test_isls_api.py.  The reference solves
one trajectory per call (isls/isls.py:379-501, isls/sls.py:319-454); these are the batched counterparts of those calls."""
import os
import sys

import numpy as np
import pytest

import isls_problems as P

pytestmark = pytest.mark.gpu


def _nominals(cfg, sel):
    """(x_nom, u_nom) of the trajectories `sel`: open-loop rollout of u0 from x0, vectorised over the batch (the per-trajectory
    python loop of isls_problems.initial_nominal would take minutes at 4096 x 200 steps; same arithmetic, same results)."""
    sel = np.asarray(list(sel))
    f, _ = P.model_callbacks(cfg)
    N = cfg["N"]
    u = cfg["u0"][sel]
    x = cfg["x0"][sel].copy()
    xs = np.zeros((len(sel), N, cfg["n"]))
    for t in range(N):
        xs[:, t] = x
        x = f(x, u[:, t])
    return xs, u.copy()


def _make(cfg, sel):
    import isls
    from isls import models
    sel = list(sel)
    s = isls.iSLS(cfg["n"], cfg["m"], cfg["N"], batch=len(sel))
    s.forward_model = models.Planar3R(cfg["dt"]) if cfg["model"] == P.MODEL_ARM3R else models.CarSimple(cfg["dt"])
    s.set_cost_variables(cfg["zs"], cfg["Qs"], cfg["seq"], cfg["u_std"])
    s.reset()
    s.nominal_values = _nominals(cfg, sel)
    return s


def test_vectorised_nominal_equals_the_per_trajectory_one():
    for cfg in (P.config3(batch=6, N=100, seed=0), P.config4(batch=6, N=200, seed=0)):
        xs, us = _nominals(cfg, [0, 3, 5])
        for i, b in enumerate([0, 3, 5]):
            x1, u1 = P.initial_nominal(cfg, b)
            assert np.array_equal(xs[i], x1) and np.array_equal(us[i], u1)


def test_config3_full_size_properties(monkeypatch):
    """Config 3: 3R arm, B = 4096, N = 100, n = 9, state + control boxes (the notebook's call: 5 candidates, 10 ADMM
    iterations per outer iteration)."""
    from isls import Box
    monkeypatch.setenv("ISLS_FF_NSEG", "3")                       # same feed-forward segmentation in both batch sizes
    B = 4096
    cfg = P.config3(batch=B, N=100, seed=0)
    kw = dict(project_x=Box(cfg["x_lo"], cfg["x_hi"]), project_u=Box(cfg["u_lo"], cfg["u_hi"]), max_iter=2,
              max_line_search_iter=cfg["max_line_search"], max_admm_iter=cfg["max_admm_iter"], rho_x=cfg["rho_x"],
              rho_u=cfg["rho_u"], alpha=1.0, tol=0.0)
    big = _make(cfg, range(B))
    c0 = np.array(big.cost, dtype=np.float64).copy()
    logs = big.ilqr_admm(log=True, **kw)
    sel = [0, 1, 1234, 2048, 4095]
    small = _make(cfg, sel)
    small.ilqr_admm(**kw)
    e, es = big.engine, small.engine
    for name in ("xhat", "uhat", "K", "k", "zx", "lx", "zu", "lu", "cost"):
        assert np.array_equal(getattr(e, name)[sel].cpu().numpy(), getattr(es, name).cpu().numpy()), name
    zx, zu = e.zx.cpu().numpy(), e.zu.cpu().numpy()
    assert np.all(zx >= cfg["x_lo"]) and np.all(zx <= cfg["x_hi"]) and np.all(zu >= cfg["u_lo"]) and np.all(zu <= cfg["u_hi"])
    c1 = np.array(big.cost, dtype=np.float64)
    assert np.all(np.isfinite(c1)) and np.all(c1 < c0) and not e.status.cpu().numpy().any()
    lg = np.stack(logs)                                            # [J, B, 2] of the last outer iteration
    assert lg.shape == (cfg["max_admm_iter"], B, 2) and np.all(np.isfinite(lg)) and np.all(lg >= 0)
    assert np.median(lg[-1, :, 0]) <= np.median(lg[0, :, 0])


def test_config4_full_size_properties(monkeypatch):
    """Config 4 (per-GPU share of the 32768 cars): B = 4096, N = 200, control box + the notebook's two rotated keep-out
    rectangles on the position (project_set_convex on the device, ISLS_PROJ_SETS)."""
    from isls import Box
    pj = sys.modules["isls.projections"]
    monkeypatch.setenv("ISLS_FF_NSEG", "3")
    B = 4096
    cfg = P.config4(batch=B, N=200, seed=0)
    rho_x = np.zeros((200, 4, 4)); rho_x[:, :2, :2] = 0.1 * np.eye(2)
    cs = pj.keepout_rectangles(4, [[-7.0, -3.0], [-3.0, -7.0]], [[2.0, 1.0], [2.0, 1.0]], -np.pi / 4)
    kw = dict(project_x=cs, project_u=Box(cfg["u_lo"], cfg["u_hi"]), max_iter=2, max_line_search_iter=20, max_admm_iter=5,
              rho_x=rho_x, rho_u=cfg["rho_u"], alpha=1.0, tol=0.0)
    big = _make(cfg, range(B))
    c0 = np.array(big.cost, dtype=np.float64).copy()
    big.ilqr_admm(**kw)
    sel = [0, 7, 1000, 3000, 4095]
    small = _make(cfg, sel)
    small.ilqr_admm(**kw)
    e, es = big.engine, small.engine
    for name in ("xhat", "uhat", "K", "k", "zx", "lx", "zu", "lu", "cost"):
        assert np.array_equal(getattr(e, name)[sel].cpu().numpy(), getattr(es, name).cpu().numpy()), name
    zu = e.zu.cpu().numpy()
    assert np.all(zu >= cfg["u_lo"]) and np.all(zu <= cfg["u_hi"])
    # the consensus states keep out of both rectangles up to the inner ADMM's own stop threshold: projecting them again
    # (numpy restatement of the reference's project_set_convex, same sets) moves them by no more than that
    zx = e.zx.cpu().numpy()[sel]
    again = np.stack([cs(z.reshape(-1)).reshape(z.shape) for z in zx])
    assert np.max(np.abs(again - zx)) < 10 * cs.threshold
    c1 = np.array(big.cost, dtype=np.float64)
    assert np.all(np.isfinite(c1)) and np.median(c1) < np.median(c0) and not (e.status.cpu().numpy() & 3).any()


@pytest.mark.parametrize("nb_dim", [1, 3])
def test_config5_full_size_properties(nb_dim):
    """Config 5: SLS-ADMM with SOC chance constraints, N = 50, B = 8192 problems that differ in target, bound and variance,
    fp32 (its stated precision): batch invariance of the one-launch ADMM loop, iteration counts within the cap, and for the
    problems that met the stop rule the consensus rows satisfy both cones (psi^-1 |Sigma^1/2 y| <= u_max -+ mu'y) to the
    inner threshold."""
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    from isls.engine import kernels
    B, N, iters = 8192, 50, 50
    Linv, r_side, rr, cs = bench.config5_problem(B, nb_dim, N)
    hip = kernels()
    f32 = np.float32
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=f32)).cuda()     # noqa: E731

    def run(sel):
        sets = [{k: (dev(v[sel]) if isinstance(v, np.ndarray) and v.ndim >= 2 and v.shape[0] == B else (dev(v) if isinstance(v, np.ndarray) else v))
                 for k, v in st.items()} for st in cs.sets]
        P_ = len(sel)
        x_u = torch.zeros(P_, r_side.shape[1], r_side.shape[2], dtype=torch.float32, device="cuda")
        z = torch.zeros_like(x_u)
        it = torch.zeros(P_, dtype=torch.int32, device="cuda")
        logs = torch.full((P_, iters, 2), float("nan"), dtype=torch.float32, device="cuda")
        hip.sls_admm(dev(Linv), dev(r_side[sel]), dev(rr), sets, x_u, alpha=1.0, tol=1e-3, max_iter=iters, rho=cs.rho,
                     inner_max_iter=cs.max_iter, threshold=cs.threshold, z=z, iters=it, logs=logs, rel_tol=1e-2)
        torch.cuda.synchronize()
        return x_u.cpu().numpy(), z.cpu().numpy(), it.cpu().numpy(), logs.cpu().numpy()
    allp = np.arange(B)
    xu, z, it, lg = run(allp)
    sel = np.array([0, 1, 4097, 8191, 5000, 77])
    xu_s, z_s, it_s, lg_s = run(sel)
    assert np.array_equal(xu[sel], xu_s) and np.array_equal(z[sel], z_s) and np.array_equal(it[sel], it_s)
    assert np.array_equal(lg[sel], lg_s, equal_nan=True)
    assert np.all(it >= 1) and np.all(it <= iters) and np.all(np.isfinite(xu))
    # problems that met the first stop rule (both residuals below tol; the second rule fires on stagnation, e.g. on an
    # unreachable bound): their consensus rows satisfy both cones up to the inner ADMM's own stop threshold
    last = lg[np.arange(B), it - 1]
    done = np.flatnonzero((it < iters) & (last[:, 0] < 1e-3))
    print("config5 full size: stopped", int(np.sum(it < iters)), "primal < 1e-3", int(np.sum(last[:, 0] < 1e-3)), "both", done.size,
          "median last residuals", np.median(last, axis=0))
    assert done.size > B // 20, done.size
    worst = 0.0
    for st in cs.sets:                                            # SOC rows: w = A y + b, |w[:-1]| <= w[-1]
        A = st["A"] if st["A"].ndim == 3 else np.broadcast_to(st["A"], (B,) + st["A"].shape)
        b_ = st["b"] if st["b"].ndim == 2 else np.broadcast_to(st["b"], (B,) + st["b"].shape)
        w = np.einsum("pij,prj->pri", A[done].astype(np.float64), z[done].astype(np.float64)) + b_[done][:, None, :]
        viol = np.linalg.norm(w[..., :-1], axis=-1) - w[..., -1]
        worst = max(worst, float(np.max(viol)))
    print("worst cone violation of the consensus rows", worst)
    assert worst < cs.threshold, worst
