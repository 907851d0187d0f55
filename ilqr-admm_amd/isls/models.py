"""Built-in forward models with device implementations (HIP kernels `rollout_ls` / `linearize`).

The reference takes arbitrary Python callbacks `forward_model(x,u)` and `get_AB(x,u)` (isls/isls.py:61-66,
isls/isls_base.py:105-111); the systems its notebooks define are provided here as descriptors that select
the device implementation.  A descriptor is also callable with the reference's calling
convention (`f(x[L,n], u[L,m]) -> [L,n]` on numpy), which is what the reference's notebooks plot with.
"""
import numpy as np

from . import _capi as capi


class Model:
    model_id = None
    x_dim = u_dim = None

    def params(self):
        raise NotImplementedError

    def get_AB(self, x, u):
        """Numpy linearisation in the reference's convention: x[N,n], u[N,m] -> A[N,n,n], B[N,n,m]."""
        raise NotImplementedError


class LTI(Model):
    """x+ = A x + B u (isls/sls_base.py:49-53); A,B [n,n],[n,m] shared or [B,n,n],[B,n,m] per trajectory."""

    def __init__(self, A, B):
        self.A, self.B = np.asarray(A, dtype=np.float64), np.asarray(B, dtype=np.float64)
        self.x_dim, self.u_dim = self.A.shape[-1], self.B.shape[-1]

    def _double_integrator(self):
        """(a, b0, b1) if A = [[I, aI],[0, I]] and B = [[b0 I],[b1 I]] exactly (get_double_integrator_AB(d, 2, dt)), else None."""
        n, m = self.x_dim, self.u_dim
        if self.A.ndim != 2 or n != 2 * m:
            return None
        a, b0, b1 = self.A[0, m], self.B[0, 0], self.B[m, 0]
        eye, zero = np.eye(m), np.zeros((m, m))
        if np.array_equal(self.A, np.block([[eye, a * eye], [zero, eye]])) and np.array_equal(self.B, np.vstack([b0 * eye, b1 * eye])):
            return np.array([a, b0, b1])
        return None

    @property
    def model_id(self):
        # a double integrator is evaluated through its Kronecker structure (ISLS_MODEL_DI): same numbers, 1/6 of the work
        return capi.MODEL_DI if self._double_integrator() is not None else capi.MODEL_LTI

    def params(self):
        di = self._double_integrator()
        if di is not None:
            return di
        if self.A.ndim == 2:
            return np.concatenate([self.A.ravel(), self.B.ravel()])
        nb = self.A.shape[0]
        return np.concatenate([self.A.reshape(nb, -1), self.B.reshape(nb, -1)], axis=1)

    def __call__(self, x, u):
        return x @ self.A.T + u @ self.B.T

    def get_AB(self, x, u):
        N = x.shape[0]
        return np.broadcast_to(self.A, (N,) + self.A.shape[-2:]).copy(), np.broadcast_to(self.B, (N,) + self.B.shape[-2:]).copy()


class Planar3R(Model):
    """Planar 3R arm of notebooks/3DoF robot (state [q, qd, ee], u = qdd), unit links, closed-form FK/J."""
    model_id = capi.MODEL_ARM3R
    x_dim, u_dim = 9, 3

    def __init__(self, dt):
        self.dt = float(dt)

    def params(self):
        return np.array([self.dt])

    @staticmethod
    def fk(q):
        c = np.cumsum(q, axis=-1)
        return np.stack([np.cos(c).sum(-1), np.sin(c).sum(-1), np.zeros(q.shape[:-1])], axis=-1)

    @staticmethod
    def jacobian(q):
        c = np.cumsum(q, axis=-1)
        J = np.zeros(q.shape[:-1] + (3, 3))
        for j in range(3):
            J[..., 0, j] = -np.sin(c[..., j:]).sum(-1)
            J[..., 1, j] = np.cos(c[..., j:]).sum(-1)
        return J

    def __call__(self, x, u):
        dt = self.dt
        q = x[..., :3] + x[..., 3:6] * dt + 0.5 * u * dt ** 2
        return np.concatenate([q, x[..., 3:6] + u * dt, self.fk(q)], axis=-1)

    def get_AB(self, x, u):
        dt, N = self.dt, x.shape[0]
        A, B = np.zeros((N, 9, 9)), np.zeros((N, 9, 3))
        for j in range(3):
            A[:, j, j] = A[:, 3 + j, 3 + j] = 1.0
            A[:, j, 3 + j] = dt
            B[:, j, j], B[:, 3 + j, j] = 0.5 * dt ** 2, dt
        J = self.jacobian(x[..., :3] + x[..., 3:6] * dt + 0.5 * u * dt ** 2)
        A[:, 6:, :3], A[:, 6:, 3:6], B[:, 6:] = J, J * dt, 0.5 * J * dt ** 2
        return A, B


class CarSimple(Model):
    """Car of notebooks/Car/Iterative LQR with state constraints.ipynb cell 6: [x, y, theta, v], [omega, a]."""
    model_id = capi.MODEL_CAR
    x_dim, u_dim = 4, 2

    def __init__(self, dt):
        self.dt = float(dt)

    def params(self):
        return np.array([self.dt])

    def __call__(self, x, u):
        dt = self.dt
        return np.stack([x[..., 0] + dt * x[..., 3] * np.cos(x[..., 2]), x[..., 1] + dt * x[..., 3] * np.sin(x[..., 2]),
                         (x[..., 2] + dt * x[..., 3] * u[..., 0]) % (2 * np.pi), x[..., 3] + dt * u[..., 1]], axis=-1)

    def get_AB(self, x, u):
        dt, N = self.dt, x.shape[0]
        A, B = np.tile(np.eye(4), (N, 1, 1)), np.zeros((N, 4, 2))
        th, v = x[..., 2], x[..., 3]
        A[:, 0, 2], A[:, 1, 2] = -dt * v * np.sin(th), dt * v * np.cos(th)
        A[:, 0, 3], A[:, 1, 3], A[:, 2, 3] = dt * np.cos(th), dt * np.sin(th), dt * u[..., 0]
        B[:, 2, 0], B[:, 3, 1] = dt * v, dt
        return A, B


class TassaCar(Model):
    """Car-parking model of notebooks/Tutorial.ipynb cell 8 (Tassa et al.): [x, y, theta, v], [front wheel angle w,
    acceleration a], axle distance `dist`; the Jacobians the notebook takes from autograd are written out here."""
    model_id = capi.MODEL_TASSA
    x_dim, u_dim = 4, 2

    def __init__(self, dt, dist=2.0):
        self.dt, self.dist = float(dt), float(dist)

    def params(self):
        return np.array([self.dt, self.dist])

    def __call__(self, x, u):
        dt, d = self.dt, self.dist
        f = dt * x[..., 3]
        sw = np.sin(u[..., 0]) * f
        b = (f * np.cos(u[..., 0]) + d) - np.sqrt(d * d - sw * sw)
        return np.stack([x[..., 0] + b * np.cos(x[..., 2]), x[..., 1] + b * np.sin(x[..., 2]), x[..., 2] + np.arcsin(sw / d),
                         x[..., 3] + u[..., 1] * dt], axis=-1)

    def get_AB(self, x, u):
        dt, d, N = self.dt, self.dist, x.shape[0]
        f, sw, cw = dt * x[..., 3], np.sin(u[..., 0]), np.cos(u[..., 0])
        r = np.sqrt(d * d - (sw * f) ** 2)
        b, dbdf, dbdw = (f * cw + d) - r, cw + (sw * sw * f) / r, -f * sw + (sw * cw * f * f) / r
        st, ct = np.sin(x[..., 2]), np.cos(x[..., 2])
        A, B = np.tile(np.eye(4), (N, 1, 1)), np.zeros((N, 4, 2))
        A[:, 0, 2], A[:, 1, 2] = -b * st, b * ct
        A[:, 0, 3], A[:, 1, 3], A[:, 2, 3] = (dbdf * dt) * ct, (dbdf * dt) * st, (sw / r) * dt
        B[:, 0, 0], B[:, 1, 0], B[:, 2, 0], B[:, 3, 1] = dbdw * ct, dbdw * st, (cw * f) / r, dt
        return A, B
