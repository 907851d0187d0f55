"""Seeded problem definitions for the five BASELINE.json configs (pure numpy, no package imports).

This module is deliberately standalone: the golden-vector generator (tests/golden/make_golden.py)
imports the *reference* ``isls`` package, the tests/bench import *this repo's* ``isls`` package, and
both need the very same inputs.  Everything here is data + closed-form model maths:

* double integrator (LTI)            -- spec: reference isls/utils.py:266-276, SURVEY Appendix A.1
* via-point quadratic cost tables    -- spec: reference isls/base.py:81-89,   SURVEY Appendix A.2
* planar 3R arm (n=9, m=3)           -- spec: notebooks/3DoF robot/State and control bound constraints.ipynb
                                        cells 9-12, 22-23 (pinocchio replaced by closed-form FK/J)
* car-simple (n=4, m=2)              -- spec: notebooks/Car/Iterative LQR with state constraints.ipynb cell 6

The numpy model callbacks below follow the reference's callback conventions:
``f(x[L,n], u[L,m]) -> [L,n]`` and ``get_AB(x[N,n], u[N,m]) -> (A[N,n,n], B[N,n,m])``.
"""
from math import factorial

import numpy as np

# model ids shared with include/isls_hip.h (ISLS_MODEL_*)
MODEL_LTI = 0
MODEL_ARM3R = 1
MODEL_CAR = 2
MODEL_DI = 3


# --------------------------------------------------------------------------------------------
# Double integrator
# --------------------------------------------------------------------------------------------
def double_integrator_AB(nb_dim, nb_deriv=2, dt=0.01):
    """A = kron(A1d, I), B = kron(B1d, I); A1d[i, i+j] = dt^j/j!, B1d[nb_deriv-j] = dt^j/j!."""
    A1 = np.zeros((nb_deriv, nb_deriv))
    for j in range(nb_deriv):
        A1 += np.diag(np.full(nb_deriv - j, dt ** j / factorial(j)), j)
    B1 = np.zeros((nb_deriv, 1))
    for j in range(1, nb_deriv + 1):
        B1[nb_deriv - j, 0] = dt ** j / factorial(j)
    eye = np.eye(nb_dim)
    return np.kron(A1, eye), np.kron(B1, eye)


def lti_f(A, B):
    def f(x, u):
        return x @ A.T + u @ B.T
    return f


def lti_get_AB(A, B, N):
    An = np.broadcast_to(A, (N,) + A.shape).copy()
    Bn = np.broadcast_to(B, (N,) + B.shape).copy()

    def get_AB(x, u):
        return An, Bn
    return get_AB


def rollout_open_loop(f, x0, u):
    """x[0]=x0, x[t+1]=f(x[t],u[t]); returns x[N,n] (the last control never acts)."""
    N = u.shape[0]
    xs = np.zeros((N, x0.shape[0]))
    x = x0.copy()
    for t in range(N):
        xs[t] = x
        x = f(x[None], u[t][None])[0]
    return xs


def via_point_cost(n, N, target, Q_final):
    """Two via-points: index 0 free (Q=0) for t<N-1, index 1 = terminal target."""
    zs = np.stack([np.zeros(n), np.asarray(target, dtype=float)])
    Qs = np.stack([np.zeros((n, n)), np.asarray(Q_final, dtype=float)])
    seq = np.zeros(N, dtype=np.int32)
    seq[N - 1] = 1
    return zs, Qs, seq


# --------------------------------------------------------------------------------------------
# Config 1: 1-D double integrator LQT-ADMM (the notebook problem at N=50 / N=100)
# --------------------------------------------------------------------------------------------
def config1(N=50):
    dt = 1.0 / N
    A, B = double_integrator_AB(1, 2, dt)
    zs, Qs, seq = via_point_cost(2, N, [1.0, 0.0], np.diag([1e6, 1e6]))
    return dict(name="di1d_lqt_admm", n=2, m=1, N=N, dt=dt, A=A, B=B, zs=zs, Qs=Qs, seq=seq,
                u_std=1e-2, x0=np.zeros(2), u_lo=-5.0, u_hi=5.0, rho_u=1e-1, tol=1e-4)


# --------------------------------------------------------------------------------------------
# Config 2 (headline): 3-D double integrator, B seeded problems
# --------------------------------------------------------------------------------------------
def config2(batch, N=100, seed=0):
    rng = np.random.default_rng(seed)
    n, m, dt = 6, 3, 0.01
    A, B = double_integrator_AB(3, 2, dt)
    x0 = np.zeros((batch, n))
    x0[:, :3] = rng.uniform(-0.5, 0.5, size=(batch, 3))
    target = np.zeros((batch, n))
    target[:, :3] = rng.uniform(0.5, 1.5, size=(batch, 3))
    Qs = np.stack([np.zeros((n, n)), 1e3 * np.eye(n)])
    zs = np.zeros((batch, 2, n))
    zs[:, 1] = target
    seq = np.zeros(N, dtype=np.int32)
    seq[N - 1] = 1
    return dict(name="di3d_ilqr_admm", n=n, m=m, N=N, dt=dt, A=A, B=B, zs=zs, Qs=Qs, seq=seq,
                u_std=1e-3, x0=x0, u0=np.zeros((batch, N, m)), u_lo=-3.0, u_hi=3.0, rho_u=1e-2,
                relax=1.0, model=MODEL_LTI)


def config_generic(n, m, batch, N=40, seed=0):
    """A seeded LTI problem of ANY state / control dimension (the reference takes any, isls/base.py:11-14): a lightly coupled,
    marginally stable A = I + dt * G, B = dt * H, a terminal via-point, a box on u -- the shape of config 2 for pairs (n, m)
    the double integrators do not produce."""
    rng = np.random.default_rng(seed + 1000 * n + m)
    dt = 0.05
    A = np.eye(n) + dt * (rng.standard_normal((n, n)) / np.sqrt(n) - 0.5 * np.eye(n))
    B = dt * rng.standard_normal((n, m))
    x0 = rng.uniform(-0.5, 0.5, size=(batch, n))
    zs = np.zeros((batch, 2, n))
    zs[:, 1] = rng.uniform(0.5, 1.5, size=(batch, n))
    Qs = np.stack([np.zeros((n, n)), 1e2 * np.eye(n)])
    seq = np.zeros(N, dtype=np.int32)
    seq[N - 1] = 1
    return dict(name=f"lti_{n}_{m}", n=n, m=m, N=N, dt=dt, A=A, B=B, zs=zs, Qs=Qs, seq=seq, u_std=1e-2, x0=x0,
                u0=np.zeros((batch, N, m)), u_lo=-2.0, u_hi=2.0, rho_u=5e-2, relax=1.0, model=MODEL_LTI)


# --------------------------------------------------------------------------------------------
# Planar 3R arm, state s=[q(3), qd(3), ee(3)], u=qdd
# --------------------------------------------------------------------------------------------
def arm_fk(q):
    c = np.cumsum(q, axis=-1)
    return np.stack([np.cos(c).sum(-1), np.sin(c).sum(-1), np.zeros(q.shape[:-1])], axis=-1)


def arm_jac(q):
    """J[0,j] = -sum_{i>=j} sin c_i ; J[1,j] = sum_{i>=j} cos c_i ; J[2,:] = 0."""
    c = np.cumsum(q, axis=-1)
    s_, c_ = np.sin(c), np.cos(c)
    J = np.zeros(q.shape[:-1] + (3, 3))
    for j in range(3):
        J[..., 0, j] = -s_[..., j:].sum(-1)
        J[..., 1, j] = c_[..., j:].sum(-1)
    return J


def arm_f(dt):
    def f(x, u):
        q = x[..., :3] + x[..., 3:6] * dt + 0.5 * u * dt ** 2
        qd = x[..., 3:6] + u * dt
        return np.concatenate([q, qd, arm_fk(q)], axis=-1)
    return f


def arm_get_AB(dt, N):
    A_, B_ = double_integrator_AB(3, 2, dt)

    def get_AB(x, u):
        A = np.zeros((N, 9, 9))
        B = np.zeros((N, 9, 3))
        A[:, :6, :6] = A_
        B[:, :6] = B_
        J = arm_jac(x[..., :3] + x[..., 3:6] * dt + 0.5 * u * dt ** 2)
        A[:, 6:, :3] = J
        A[:, 6:, 3:6] = J * dt
        B[:, 6:] = 0.5 * J * dt ** 2
        return A, B
    return get_AB


def config3(batch, N=100, seed=0, q0_noise=0.1):
    rng = np.random.default_rng(seed)
    n, m, dt = 9, 3, 0.01
    q0 = np.array([np.pi / 3, -np.pi / 2, -np.pi / 4])[None] + q0_noise * rng.standard_normal((batch, 3))
    if batch >= 1:
        q0[0] = [np.pi / 3, -np.pi / 2, -np.pi / 4]          # trajectory 0 is the notebook problem itself
    x0 = np.concatenate([q0, np.zeros((batch, 3)), arm_fk(q0)], axis=-1)
    target = np.array([0, 0, 0, 0, 0, 0, 1.5, 1.0, 0.0])
    Q_final = np.diag([0, 0, 0, 1e6, 1e6, 1e6, 0, 1e6, 0.0])
    zs, Qs, seq = via_point_cost(n, N, target, Q_final)
    Qr = np.zeros((N, n, n))
    Qr[-1, 6, 6] = 1e1
    Qr[:, 3:6, 3:6] = np.eye(3) * 1e-2
    x_lo = np.full((N, n), -np.inf)
    x_hi = np.full((N, n), np.inf)
    x_lo[:, 3:6], x_hi[:, 3:6] = -1.5, 1.5
    x_lo[-1, 6], x_hi[-1, 6] = 0.5, 1.0
    return dict(name="arm3r_ilqr_admm", n=n, m=m, N=N, dt=dt, zs=zs, Qs=Qs, seq=seq, u_std=1e-4,
                x0=x0, u0=np.ones((batch, N, m)), u_lo=-6.0, u_hi=6.0, x_lo=x_lo, x_hi=x_hi,
                rho_x=Qr, rho_u=1e-3, relax=1.0, model=MODEL_ARM3R, max_admm_iter=10, max_line_search=5)


# --------------------------------------------------------------------------------------------
# Car-simple, state [x, y, theta, v], control [omega, a]
# --------------------------------------------------------------------------------------------
def car_f(dt):
    def f(x, u):
        x1 = x[..., 0] + dt * x[..., 3] * np.cos(x[..., 2])
        x2 = x[..., 1] + dt * x[..., 3] * np.sin(x[..., 2])
        x3 = (x[..., 2] + dt * x[..., 3] * u[..., 0]) % (2 * np.pi)
        x4 = x[..., 3] + dt * u[..., 1]
        return np.stack([x1, x2, x3, x4], axis=-1)
    return f


def car_get_AB(dt, N):
    def get_AB(x, u):
        A = np.zeros((N, 4, 4))
        B = np.zeros((N, 4, 2))
        A[:] = np.eye(4)
        th, v = x[..., 2], x[..., 3]
        A[:, 0, 2] = -dt * v * np.sin(th)
        A[:, 1, 2] = dt * v * np.cos(th)
        A[:, 0, 3] = dt * np.cos(th)
        A[:, 1, 3] = dt * np.sin(th)
        A[:, 2, 3] = dt * u[..., 0]
        B[:, 2, 0] = dt * v
        B[:, 3, 1] = dt
        return A, B
    return get_AB


def config4(batch, N=200, seed=0, x0_noise=0.05):
    rng = np.random.default_rng(seed)
    n, m, dt = 4, 2, 0.03
    x0 = np.array([0.0, -2.0, np.pi / 2, 0.0])[None] + x0_noise * rng.standard_normal((batch, n))
    x0[:, 3] = 0.0
    target = np.array([-5.0, -5.0, np.pi / 4, 0.0])
    zs, Qs, seq = via_point_cost(n, N, target, 1e2 * np.eye(n))
    u0 = 0.01 * rng.standard_normal((batch, N, m))
    # state box used as the state constraint of this config (velocity limit + workspace box)
    x_lo = np.full((N, n), -np.inf)
    x_hi = np.full((N, n), np.inf)
    x_lo[:, 3], x_hi[:, 3] = -2.0, 2.0
    x_lo[:, 0], x_hi[:, 0] = -6.0, 1.0
    Qr = np.zeros((N, n, n))
    Qr[:, 0, 0] = 1e-1
    Qr[:, 3, 3] = 1e-1
    return dict(name="car_ilqr_admm", n=n, m=m, N=N, dt=dt, zs=zs, Qs=Qs, seq=seq, u_std=1e-2,
                x0=x0, u0=u0, u_lo=-0.5, u_hi=0.5, x_lo=x_lo, x_hi=x_hi, rho_x=Qr, rho_u=1e1,
                relax=1.0, model=MODEL_CAR, max_admm_iter=5, max_line_search=20)


def initial_nominal(cfg, b):
    """Open-loop rollout of trajectory ``b`` of a batched config from its x0 with its initial controls."""
    if cfg["model"] == MODEL_LTI:
        f = lti_f(cfg["A"], cfg["B"])
    elif cfg["model"] == MODEL_ARM3R:
        f = arm_f(cfg["dt"])
    else:
        f = car_f(cfg["dt"])
    u = cfg["u0"][b]
    return rollout_open_loop(f, cfg["x0"][b], u), u.copy()


def model_callbacks(cfg):
    """(f, get_AB) numpy callbacks of a config, in the reference's calling convention."""
    N, dt = cfg["N"], cfg["dt"]
    if cfg["model"] == MODEL_LTI:
        return lti_f(cfg["A"], cfg["B"]), lti_get_AB(cfg["A"], cfg["B"], N)
    if cfg["model"] == MODEL_ARM3R:
        return arm_f(dt), arm_get_AB(dt, N)
    return car_f(dt), car_get_AB(dt, N)
