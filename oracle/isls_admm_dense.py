"""Dense (batch-form) CPU restatement of `iSLS.isls_admm` -- TEST INFRASTRUCTURE ONLY (see isls_oracle_impl.h).

Follows isls/isls.py:503-712 with the dense transfer matrices of `Base.AB` (isls/base.py:98-119), one problem at a
time in numpy (small cases only: an (N m)^2 inverse per outer iteration).  The nonlinear pieces are the C oracle's:
linearisation, open-loop rollouts + cost + first arg-min (`oracle_rollout_ls` with zero gains restates
`rollout_batch` + `cost_function` + `np.argmin`, isls.py:593-599), the nominal's cost (`oracle_expand_quadratic`) and the
row projection (`oracle_project_rows` = project_set_convex).  Pinned by tests/golden/g9_isls_admm.npz, produced by the
reference itself on the 3R arm (tests/test_oracle_golden.py).  Quadratic via-point cost only (`get_Cs=None` branch).
"""
import numpy as np

ALPHAS = 10.0 ** np.linspace(0.0, -5.0, 50)                  # isls/isls_base.py:10-11


def transfer_matrices(A, B):
    """Sw [N n, N n], Su [N n, N m] of the linearisation A [N,n,n], B [N,n,m]: column block i-1 is column block i times
    A[i-1] (resp. B[i-1]) on the rows below, filled from the last column to the first (isls/base.py:112-119)."""
    N, n, m = A.shape[0], A.shape[1], B.shape[2]
    Sw, Su = np.eye(N * n), np.zeros((N * n, N * m))
    for i in range(N - 1, 0, -1):
        below = Sw[i * n:, i * n:(i + 1) * n]
        Su[i * n:, (i - 1) * m:i * m] = below @ B[i - 1]
        Sw[i * n:, (i - 1) * n:i * n] = below @ A[i - 1]
    return Sw, Su


def block_diag(blocks):
    N, d = blocks.shape[0], blocks.shape[1]
    out = np.zeros((N * d, N * d))
    for t in range(N):
        out[t * d:(t + 1) * d, t * d:(t + 1) * d] = blocks[t]
    return out


class DenseIslsAdmm:
    """One problem.  kern: the oracle `Kernels`; pa: arrays of tests/helpers.problem_arrays for ONE trajectory (B == 1)."""

    def __init__(self, kern, pa, dim, project_u=None, project_x=None, rho_u=None, rho_x=None, relax=1.0, threshold=1e-3):
        assert pa["B"] == 1
        self.kern, self.pa, self.dim = kern, pa, int(dim)
        self.N, self.n, self.m = pa["N"], pa["n"], pa["m"]
        self.project_u, self.project_x = project_u, project_x                  # callables (rows, nominal) -> rows, or None
        N, n, m = self.N, self.n, self.m
        self.Rr = None if project_u is None else block_diag(np.broadcast_to(rho_u, (N, m, m)) if np.ndim(rho_u) else np.tile(rho_u * np.eye(m), (N, 1, 1)))
        self.Qr = None if project_x is None else block_diag(np.broadcast_to(rho_x, (N, n, n)) if np.ndim(rho_x) else np.tile(rho_x * np.eye(n), (N, 1, 1)))
        self.relax, self.threshold = relax, threshold
        self.x_nom, self.u_nom = pa["xhat"][0].astype(np.float64).copy(), pa["uhat"][0].astype(np.float64).copy()
        self.Q = block_diag(pa["Qtab"][pa["seq"]])                               # find_precs, isls/base.py:87
        self.R = pa["u_std"] * np.eye(N * m)
        self.xd = pa["ztab"].reshape(-1, n)[pa["seq"]].reshape(-1)              # find_mus, isls/base.py:88
        self.cost = self.nominal_cost()
        self.cost_log = [self.cost]
        self.trace = []

    def nominal_cost(self):
        pa, c = self.pa, np.zeros(1)
        z = np.zeros
        self.kern.expand_quadratic(pa["Qtab"], pa["ztab"], pa["seq"], pa["u_std"], z((1, self.N, self.n)), z((1, self.N, self.m)),
                                   xhat=self.x_nom[None].copy(), uhat=self.u_nom[None].copy(), cost=c)
        return float(c[0])

    def line_search(self, delta_u, L):
        """rollout_batch(x_nom, u_nom + alpha delta_u) for alphas[:L], cost_function, first arg-min (isls.py:593-599)."""
        pa, N, n, m = self.pa, self.N, self.n, self.m
        xs, us = np.zeros((1, N, n)), np.zeros((1, N, m))
        best, cost_new = np.zeros(1, dtype=np.int32), np.zeros(1)
        self.kern.rollout_ls(pa["model"], pa["model_par"], np.zeros((1, N, m, n)), np.ascontiguousarray(delta_u.reshape(1, N, m)),
                             self.x_nom[None].copy(), self.u_nom[None].copy(), ALPHAS[:L].copy(), pa["Qtab"], pa["ztab"],
                             pa["seq"], pa["u_std"], xs, us, best=best, cost_new=cost_new)
        return int(best[0]), xs[0]

    def outer_iteration(self, max_admm_iter, L, z_x, z_u):
        pa, N, n, m, dim = self.pa, self.N, self.n, self.m, self.dim
        A, B = np.zeros((1, N, n, n)), np.zeros((1, N, n, m))
        self.kern.linearize(pa["model"], pa["model_par"], self.x_nom[None].copy(), self.u_nom[None].copy(), A, B)
        Sw, Su = transfer_matrices(A[0], B[0])
        self.Sw, self.Su = Sw, Su                                               # what `controller` reads afterwards
        Sx = Sw[:, :dim]
        # quadratic-cost branch, isls.py:560-566
        xd, ud = self.xd - self.x_nom.reshape(-1), -self.u_nom.reshape(-1)
        SuTQ = Su.T @ self.Q
        l_side = SuTQ @ Su + self.R
        r_ff = SuTQ @ xd + self.R @ ud
        r_fb = -SuTQ @ Sx
        if self.project_x is not None:                                          # isls.py:568-573
            SuTQr = Su.T @ self.Qr
            l_side = l_side + SuTQr @ Su
            r_fb = r_fb - SuTQr @ Sx
        if self.project_u is not None:
            l_side = l_side + self.Rr
        l_inv = np.linalg.inv(l_side)
        r_side = np.concatenate([r_ff[:, None], r_fb], axis=1)

        def f_argmin(reg_x, reg_u):                                             # isls.py:578-611
            rhs = r_side.copy()
            if self.project_x is not None:
                rhs = rhs + SuTQr @ reg_x
            if self.project_u is not None:
                rhs = rhs + self.Rr @ reg_u
            du = l_inv @ rhs
            dx = Su @ du
            dx[:, 1:] += Sx
            ind, x_best = self.line_search(du[:, 0], L)
            du[:, 0] = (du[:, 0].reshape(N, m) * ALPHAS[ind]).reshape(-1)
            dx[:, 0] = (x_best - self.x_nom).reshape(-1)
            return dx, du

        lmb_x = lmb_u = 0.0
        prim = dual = 1e6
        logs = []
        for _ in range(max_admm_iter):                                          # isls.py:619-680
            x_x, x_u = f_argmin(z_x - lmb_x if self.project_x is not None else None,
                                z_u - lmb_u if self.project_u is not None else None)
            prev_prim, prev_dual = prim, dual
            prim = dual = 0.0
            if self.project_x is not None:
                z_prev = z_x
                z_x = self.project_x(self.relax * x_x + (1 - self.relax) * z_x + lmb_x, self.x_nom)
                r = x_x - z_x
                lmb_x = lmb_x + r
                dual += np.linalg.norm(self.Qr @ (z_x - z_prev))
                prim += np.linalg.norm(self.Qr @ r)
            if self.project_u is not None:
                z_prev = z_u
                z_u = self.project_u(self.relax * x_u + (1 - self.relax) * z_u + lmb_u, self.u_nom)
                r = x_u - z_u
                lmb_u = lmb_u + r
                dual += np.linalg.norm(self.Rr @ (z_u - z_prev))
                prim += np.linalg.norm(self.Rr @ r)
            logs.append((prim, dual))
            self.trace.append(dict(x_x=x_x.copy(), x_u=x_u.copy(), z_u=None if self.project_u is None else z_u.copy()))
            if prim < self.threshold and dual < self.threshold:
                break
            if (abs(prev_prim - prim) / (prev_prim + 1e-30) < 1e-3 and abs(prev_dual - dual) / (prev_dual + 1e-30) < 1e-3):
                break
        self.u_nom = self.u_nom + x_u[:, 0].reshape(N, m)                       # isls.py:684-687
        self.x_nom = self.x_nom + x_x[:, 0].reshape(N, n)
        self.cost = self.nominal_cost()
        self.cost_log.append(self.cost)
        return x_x, x_u, z_x, z_u, logs

    def solve(self, k_max, max_admm_iter, L):
        N, n, m, C = self.N, self.n, self.m, self.dim + 1
        z_x, z_u = np.zeros((N * n, C)), np.zeros((N * m, C))
        self.logs = []
        for _ in range(k_max):
            prev = self.cost
            x_x, x_u, z_x, z_u, logs = self.outer_iteration(max_admm_iter, L, z_x, z_u)
            self.logs.append(logs)
            if abs(self.cost - prev) < 1e-4:                                    # isls.py:695-697
                break
            if len(self.cost_log) >= 5 and abs(np.mean(self.cost_log[-4:]) - np.mean(self.cost_log[-8:-4])) < 1e-3:
                break
        self.x_x, self.x_u = x_x, x_u
        return x_u[:, 0], x_u[:, 1:]


def controller(Sw, Su, phi_u, du, n):
    """K, k of SLS.controller (isls/sls.py:235-242) for feedback on the first columns only: PHI_U = [phi_u, 0]."""
    PHI_U = np.zeros((Su.shape[1], Sw.shape[1]))
    PHI_U[:, :phi_u.shape[1]] = phi_u
    K = PHI_U @ np.linalg.inv(Sw + Su @ PHI_U)
    return K, (np.eye(Su.shape[1]) - K @ Su) @ du


def shifted_sets_projection(kern, cs):
    """The notebook's project_u(rows, nominal) for a ConvexSets over the rows [nominal + d, phi] (cell 25 of
    notebooks/3DoF robot/State bounds and robust control bounds.ipynb): add the nominal to column 0, run
    project_set_convex over all rows (oracle_project_rows), take it off again."""
    sets = [{k: (np.ascontiguousarray(v, dtype=np.float64) if isinstance(v, np.ndarray) else v) for k, v in st.items()} for st in cs.sets]

    def project(rows, nominal):
        y = np.ascontiguousarray(rows, dtype=np.float64).copy()
        y[:, 0] += nominal.reshape(-1)
        y3 = y[None].copy()
        kern.project_rows(y3, y3, sets, rho=cs.rho, max_iter=cs.max_iter, threshold=cs.threshold)
        y = y3[0]
        y[:, 0] -= nominal.reshape(-1)
        return y
    return project
