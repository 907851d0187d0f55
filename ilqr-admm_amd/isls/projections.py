"""Projection operators of the ADMM z-step (reference: isls/projections.py).

Two kinds of objects live here:

* `Box` -- a *descriptor* the HIP kernel `admm_update` executes on the device (ISLS_PROJ_BOX); it is also a
  plain callable on numpy vectors so that it can be passed wherever the reference expects `project_x` /
  `project_u` callbacks.
* numpy functions with the reference's names and semantics (`project_bound`, `project_soc_unit`,
  `project_square_batch`, `project_set_convex`, ...).  They are used by the generic (callback) ADMM path and
  are pinned against reference outputs in tests/test_projections.py.  Only the box has a device kernel in
  this round (the others are "next").
"""
import numpy as np


class Box:
    """l <= P(x) <= u, element-wise (np.clip; isls/projections.py:7-11).  lo/hi: scalars, [d], [N,d] or flat [N*d]."""

    def __init__(self, lo, hi):
        self.lo, self.hi = lo, hi

    def __call__(self, x):
        """x: anything whose last axis is d (or a flat [N*d] vector of N rows, as the ADMM callbacks receive it)."""
        x = np.asarray(x)

        def fit(v):
            v = np.asarray(v, dtype=np.float64)
            if v.ndim == 0 or v.shape == x.shape:
                return v
            if v.size == x.size:
                return v.reshape(x.shape)
            if x.ndim == 1 and v.ndim == 1 and x.size % v.size == 0:      # per-dimension bounds on a flat [N*d] vector
                return np.tile(v, x.size // v.size)
            return v
        return np.clip(x, fit(self.lo), fit(self.hi))

    def bounds(self, N, d):
        """(lo, hi) as float64 arrays broadcastable to [N, d]."""
        def expand(v):
            v = np.asarray(v, dtype=np.float64)
            if v.ndim == 0:
                return np.full((1, d), float(v))
            if v.size == N * d:
                return v.reshape(N, d)
            return v.reshape(1, d)
        return expand(self.lo), expand(self.hi)


SET_BOX, SET_SOC_UNIT, SET_SQUARE, SET_LINEAR, SET_QUADRATIC, SET_SHELL, SET_MULTILINEAR = 1, 2, 3, 4, 5, 6, 7   # include/isls_hip.h ISLS_SET_*
ALG_ADMM, ALG_DYKSTRA, ALG_SOC = 0, 1, 2                                           # ISLS_PROJ_ALG_*


class ConvexSets:
    """Intersection of convex(ish) sets acting on the coordinate block [c0, c0+d) of every time-step row of a flat
    [N*dim] vector, evaluated with `project_set_convex` (isls/projections.py:289-374) over the N rows of one call.

    It is what the reference's notebooks assemble by hand from `project_set_convex`, `project_soc_unit` and
    `project_square_batch` closures (Car "state constraints" notebook cell 18, Double-integrator "control bounds"
    cell 15).  Calling it on a numpy vector runs the numpy operators below (host route); the solvers recognise the
    object and run the same iteration on the device instead (ISLS_PROJ_SETS / isls_project_rows).

    sets: list of dict(kind=SET_*, dim=, A=[dim,d], b=[dim], par=[...]) -- `par` as documented in include/isls_hip.h.
    """

    def __init__(self, dim, cols, sets, rho=1.0, max_iter=200, threshold=1e-4, algorithm=ALG_ADMM, rows=None, then=None):
        """algorithm: ALG_ADMM = project_set_convex, ALG_DYKSTRA = project_set_convex_dykstra (sets act on the rows themselves;
        threshold is its `tol`), ALG_SOC = project_soc (one SOC set with A, b).  rows: indices (or a boolean mask) of the
        time-step rows the sets apply to, the others pass through (None = all).  then: another ConvexSets applied to the
        result (the obstacle notebook chains project_set_convex and Dykstra; one stage per group of rows otherwise)."""
        self.dim, self.cols, self.sets = int(dim), (int(cols[0]), int(cols[1])), list(sets)
        self.rho, self.max_iter, self.threshold = float(rho), int(max_iter), float(threshold)
        self.algorithm, self.rows, self.then = int(algorithm), rows, then

    def row_mask(self, R):
        """int32 [R] mask of the rows this stage touches, or None for all rows."""
        if self.rows is None:
            return None
        m = np.zeros(R, dtype=np.int32)
        m[np.asarray(self.rows)] = 1
        return m

    def stages(self):
        out, cur = [], self
        while cur is not None:
            out.append(cur)
            cur = cur.then
        return out

    @staticmethod
    def _primitive(st):
        kind, par = st["kind"], st.get("par")
        if kind == SET_BOX:
            d = st["dim"]
            return lambda v: project_bound(v, par[:d], par[d:2 * d])
        if kind == SET_SOC_UNIT:
            return project_soc_unit
        if kind == SET_LINEAR:
            return lambda v: project_linear_batch(v, np.broadcast_to(par[2:], np.shape(v)), par[0], par[1])
        if kind == SET_QUADRATIC:
            return lambda v: project_quadratic_batch(v, par[0], par[1])
        if kind == SET_SHELL:
            return lambda v: project_quadratic_batch(v - par[2:], par[0], par[1]) + par[2:]
        if kind == SET_MULTILINEAR:
            q = int(par[0])
            l, u, M = par[1:1 + q], par[1 + q:1 + 2 * q], par[1 + 2 * q:].reshape(q, -1)
            return lambda v: np.stack([project_multilinear(r, M, l, u) for r in np.atleast_2d(v)])
        q = int(par[0])
        l, u, c = par[1], par[2], par[3:3 + q]
        W, Wi = par[3 + q:3 + q + q * q].reshape(q, q), par[3 + q + q * q:3 + q + 2 * q * q].reshape(q, q)

        def square(v):
            out = np.array(v, dtype=np.float64, copy=True)
            out[:, :q] = project_square_batch((out[:, :q] - c) @ W.T, l, u) @ Wi.T + c
            return out
        return square

    def __call__(self, flat):
        y = np.array(flat, dtype=np.float64, copy=True).reshape(-1, self.dim)
        c0, d = self.cols
        sel = slice(None) if self.rows is None else np.flatnonzero(self.row_mask(y.shape[0]))
        arg = y[sel, c0:c0 + d].copy()
        prims = [self._primitive(s) for s in self.sets]
        if self.algorithm == ALG_DYKSTRA:
            blk = project_set_convex_dykstra(arg, prims, max_iter=self.max_iter, tol=self.threshold)
        elif self.algorithm == ALG_SOC:
            blk = project_soc(arg, np.asarray(self.sets[0]["A"], dtype=np.float64), np.asarray(self.sets[0]["b"], dtype=np.float64),
                              rho=self.rho, max_iter=self.max_iter, tol=self.threshold)
        else:
            blk = project_set_convex(arg, [np.asarray(s["A"], dtype=np.float64) for s in self.sets],
                                     [np.asarray(s["b"], dtype=np.float64) for s in self.sets], projections=prims, rho=self.rho,
                                     max_iter=self.max_iter, threshold=self.threshold)
        y[sel, c0:c0 + d] = np.atleast_2d(blk)
        out = y.reshape(np.shape(flat))
        return self.then(out) if self.then is not None else out


def keepout_rectangles(dim, centres, sizes, angle, margin=0.5, upper=1e5, rho=10.0, max_iter=15, threshold=1e-3):
    """State constraint of notebooks/Car/Iterative LQR with state constraints.ipynb cell 18: stay outside rotated
    rectangles (width, length = sizes[i] + margin, rotated by `angle`, centred at centres[i]) in the first two
    coordinates of a `dim`-vector."""
    Rm = np.array([[np.cos(angle), -np.sin(angle)], [np.sin(angle), np.cos(angle)]])
    sets = []
    for c, a in zip(np.asarray(centres, dtype=np.float64), np.asarray(sizes, dtype=np.float64)):
        a_safe = a + margin
        W = np.diag(a_safe[0] / a_safe) @ Rm.T
        par = np.concatenate([[2, a_safe[0] / 2, upper], c, W.ravel(), np.linalg.inv(W).ravel()])
        sets.append(dict(kind=SET_SQUARE, dim=dim, A=np.eye(dim), b=np.zeros(dim), par=par))
    return ConvexSets(dim, (0, dim), sets, rho=rho, max_iter=max_iter, threshold=threshold)


def spherical_keepout(dim, centres, radii, margin=1.1, upper=1e2, q=None, admm_iter=5, admm_threshold=1e-2, dykstra_iter=50,
                      dykstra_tol=1e-5):
    """State constraint of notebooks/Double integrator/LQR and SLS with spherical obstacle avoidance.ipynb cell 12: keep the
    first q coordinates (default: all) of every row outside the balls |y - c_i| < margin r_i -- project_set_convex over the
    shells (margin r_i)^2 / 2 <= |y - c_i|^2 / 2 <= upper, then project_set_convex_dykstra over the same shells, both on the
    device (ISLS_SET_SHELL, ISLS_PROJ_ALG_ADMM then ISLS_PROJ_ALG_DYKSTRA)."""
    q = dim if q is None else int(q)
    mk = lambda: [dict(kind=SET_SHELL, dim=q, A=np.eye(q), b=np.zeros(q),                       # noqa: E731
                       par=np.concatenate([[0.5 * (margin * r) ** 2, upper], np.asarray(c, dtype=np.float64)]))
                  for c, r in zip(centres, radii)]
    dyk = ConvexSets(dim, (0, q), mk(), max_iter=dykstra_iter, threshold=dykstra_tol, algorithm=ALG_DYKSTRA)
    return ConvexSets(dim, (0, q), mk(), rho=1.0, max_iter=admm_iter, threshold=admm_threshold, then=dyk)


def chance_constraint_rows(p, upper, lower, var_x0, psi_inv, x0_pos=None, rho=10.0, max_iter=100, threshold=1e-3):
    """Chance constraints lower <= u <= upper on the rows y = [d_u, phi_u] in R^(1+p) of the SLS variable, robust to
    x0_pos ~ N(x0_pos, var_x0 I) with confidence Psi(psi_inv): two unit-SOC images
        psi_inv |Sigma^(1/2) y| <= upper - mu'y ,  psi_inv |Sigma^(1/2) y| <= mu'y - lower ,  mu = [1, x0_pos]
    (notebooks/Double integrator/LQR and SLS with control bounds.ipynb cell 15).  upper / lower / var_x0 / psi_inv may be
    arrays of length B for B problems."""
    up, lo, var, psi = (np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in (upper, lower, var_x0, psi_inv))
    B = max(len(up), len(lo), len(var), len(psi))
    up, lo, var, psi = (np.broadcast_to(v, (B,)) for v in (up, lo, var, psi))
    mu = np.zeros(p + 1)
    mu[0] = 1.0
    if x0_pos is not None:
        mu[1:] = x0_pos
    A0, A1, b0, b1 = [], [], [], []
    for b in range(B):
        Au = np.diag(np.sqrt(np.concatenate([[0.0], np.full(p, var[b])])))
        A0.append(np.concatenate([Au, (-mu / psi[b])[None]], 0)), A1.append(np.concatenate([Au, (mu / psi[b])[None]], 0))
        b0.append(np.append(np.zeros(p + 1), up[b] / psi[b])), b1.append(np.append(np.zeros(p + 1), -lo[b] / psi[b]))
    sq = (lambda a: np.ascontiguousarray(a[0])) if B == 1 else (lambda a: np.ascontiguousarray(np.stack(a)))
    sets = [dict(kind=SET_SOC_UNIT, dim=p + 2, A=sq(A0), b=sq(b0)), dict(kind=SET_SOC_UNIT, dim=p + 2, A=sq(A1), b=sq(b1))]
    return ConvexSets(p + 1, (0, p + 1), sets, rho=rho, max_iter=max_iter, threshold=threshold)


def identify_box(project, size, rng_seed=0):
    """Recognise an opaque `project(flat_vector)` callback as a box and return the equivalent `Box`, else None.

    The reference's notebooks pass lambdas such as `lambda u: project_bound(u, -5, 5)`.  A map P is the
    projection on a box iff P(v) = clip(v, lo, hi) with lo = P(-inf..), hi = P(+inf..); we read lo/hi off two
    extreme probes and verify the identity on random vectors (exact comparison).
    """
    big = 1e300
    try:
        hi = np.asarray(project(np.full(size, big)), dtype=np.float64).reshape(-1)
        lo = np.asarray(project(np.full(size, -big)), dtype=np.float64).reshape(-1)
    except Exception:
        return None
    if hi.shape != (size,) or lo.shape != (size,) or np.any(lo > hi):
        return None
    hi = np.where(hi >= big, np.inf, hi)
    lo = np.where(lo <= -big, -np.inf, lo)
    rng = np.random.default_rng(rng_seed)
    fin = np.isfinite(lo) | np.isfinite(hi)
    scale = 1.0 + (np.max(np.abs(np.concatenate([lo[np.isfinite(lo)], hi[np.isfinite(hi)]]))) if fin.any() else 0.0)
    for _ in range(4):
        v = rng.standard_normal(size) * 3.0 * scale
        try:
            out = np.asarray(project(v.copy()), dtype=np.float64).reshape(-1)
        except Exception:
            return None
        if out.shape != (size,) or not np.array_equal(out, np.clip(v, lo, hi)):
            return None
    return Box(lo, hi)


# ---- primitives ------------------------------------------------------------------------------------
def project_bound(x, l, u):
    """l <= P(x) <= u  (isls/projections.py:7-11)."""
    return np.clip(x, l, u)


def project_linear_batch(x, a, l, u):
    """Rows x_i onto the slab l <= a_i'x_i <= u  (isls/projections.py:30-43)."""
    ax = np.einsum("ij,ij->i", x, a)
    aa = np.einsum("ij,ij->i", a, a) + 1e-30
    excess = np.where(ax > u, ax - u, np.where(ax < l, ax - l, 0.0))
    return x - (excess / aa)[:, None] * a


def project_linear(x, a, l, u):
    """x onto l <= a'x <= u  (isls/projections.py:13-28)."""
    if x.ndim == 2:
        return project_linear_batch(x, a, l, u)
    ax, aa = a.dot(x), a.dot(a) + 1e-30
    mu = ax - u if ax > u else (ax - l if ax < l else 0.0)
    return x - mu * a / aa


def project_affine(x, a, b, l, u):
    """l <= a'x + b <= u  (isls/projections.py:64-68)."""
    return project_linear(x, a, l - b, u - b)


def project_multilinear(x, A, l, u):
    """Boundary projection for l <= A x <= u (not the minimum-norm one; isls/projections.py:46-61)."""
    Ax = A.dot(x)
    target = np.where(Ax > u, u, np.where(Ax < l, l, Ax))
    return x - A.T @ (np.linalg.inv(A @ A.T) @ (Ax - target))


def project_quadratic_batch(x, l, u):
    """Rows onto the shell l <= 0.5 |x|^2 <= u  (isls/projections.py:91-105)."""
    val = 0.5 * np.sum(x * x, axis=-1)
    nrm = np.linalg.norm(x, axis=-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        out_hi = x * (np.sqrt(2 * u) / nrm)[:, None]
        out_lo = x * (np.sqrt(2 * l) / nrm)[:, None]
    z = np.where((val > u)[:, None], out_hi, x)
    return np.where((l > val)[:, None], out_lo, z)


def project_quadratic(x, l, u):
    if x.ndim == 2:
        return project_quadratic_batch(x, l, u)
    raise NotImplementedError                                  # as in the reference (isls/projections.py:77)


def project_quadratic_b(x, b, l, u):
    """l <= 0.5 x'x + b'x <= u  (isls/projections.py:107-115)."""
    const = 0.5 * b.T.dot(b)
    return project_quadratic(x + b, l + const, u + const) - b


def project_soc_unit_batch(z, t):
    """Rows (z_i, t_i) onto the second-order cone |z| <= t.  Mask precedence follows the reference
    (isls/projections.py:140-162): boundary formula, then the polar-cone zeroing, then the interior identity."""
    nz = np.linalg.norm(z, axis=-1)
    half = (nz + t) / 2
    inside = nz <= t
    polar = (nz <= -t) | (t < 0)
    z_out = np.where(inside[:, None], z, np.where(polar[:, None], 0.0, half[:, None] * z / (nz[:, None] + 1e-30)))
    t_out = np.where(inside, t, np.where(polar, 0.0, half))
    return z_out, t_out


def project_soc_unit(zt):
    """[z, t] onto |z| <= t; accepts a vector or rows  (isls/projections.py:118-137)."""
    z, t = zt[..., :-1], zt[..., -1]
    if z.ndim == 2:
        z_, t_ = project_soc_unit_batch(z, t)
        return np.concatenate([z_, t_[:, None]], axis=1)
    nz = np.linalg.norm(z)
    if nz <= t:
        return np.append(z, t)
    if nz <= -t:
        return np.append(z * 0, t * 0)
    half = (nz + t) / 2
    return np.append(half * z / (nz + 1e-30), half)


def project_unit_ball(x):
    n = np.linalg.norm(x)
    return x if n <= 1 else x / n


def project_square_batch(x, l, u):
    """Rows onto l <= |x|_inf <= u  (isls/projections.py:256-266): the dominant coordinate is pushed out to l."""
    z = np.array(x, dtype=float, copy=True)
    j = np.argmax(np.abs(x), axis=-1)
    rows = np.nonzero(np.abs(x).max(axis=-1) < l)[0]
    z[rows, j[rows]] = l * np.sign(x[rows, j[rows]])
    return np.clip(z, -u, u)


def project_square(x, l, u):
    """Vector version (isls/projections.py:245-254)."""
    z = np.array(x, dtype=float, copy=True)
    j = int(np.argmax(np.abs(x)))
    if abs(x[j]) < l:
        z[j] = l * np.sign(x[j])
    return np.clip(z, -u, u)


def project_square_c(x, c, l, u):
    return project_square(x - c, l, u) + c


def project_block_lower_triangular(z, x_dim, u_dim, N):
    """isls/projections.py:277-282."""
    for i in range(N):
        z[i * u_dim, i * x_dim:(i + 1) * x_dim] = 0.0
    return z


projections = {"SOC": project_soc_unit, "bound": project_bound, "linear": project_linear,
               "quadratic": project_quadratic, "square": project_square}


def _consensus_residual_loop(update, max_iter, threshold, rel=1e-5):
    """Shared stop logic of the reference's inner ADMM solvers: max-norm residuals below `threshold`, or both
    relative changes below 1e-5 (isls/projections.py:349-372)."""
    prev_p, prev_d = 1e5, 1e5
    for j in range(max_iter):
        p, d = update()
        if p < threshold and d < threshold:
            break
        if j < max_iter - 1:
            if abs(prev_p - p) / (prev_p + 1e-30) < rel and abs(prev_d - d) / (prev_d + 1e-30) < rel:
                break
        prev_p, prev_d = p, d


def project_set_convex(x0, As=[], bs=[], projections=[], rho=1, max_iter=200, threshold=1e-4, verbose=False):
    """Projection onto the intersection {x : A_i x + b_i in C_i} by consensus ADMM (isls/projections.py:289-374):
    x = (I + rho sum A_i'A_i)^-1 (x0 + rho sum A_i'(z_i - b_i - lmb_i)); z_i = P_i(A_i x + b_i + lmb_i); lmb_i += A_i x + b_i - z_i."""
    X0 = (x0[None] if x0.ndim == 1 else x0).T
    single = X0.shape[1] == 1                 # the reference squeezes whenever nb_size == 1, also for a [1, d] input (projections.py:371-374)
    x = X0.copy()
    z = [A @ x + b[:, None] for A, b in zip(As, bs)]
    lmb = [np.zeros_like(zi) for zi in z]
    lhs_inv = np.linalg.inv(np.eye(X0.shape[0]) + rho * sum(A.T @ A for A in As))
    state = {"x": x}

    def update():
        rhs = sum(A.T @ (zi - b[:, None] - li) for A, b, zi, li in zip(As, bs, z, lmb))
        xk = lhs_inv @ (X0 + rho * rhs)
        p_max = d_max = 0.0
        for i, (A, b, P) in enumerate(zip(As, bs, projections)):
            Axb = A @ xk + b[:, None]
            z_new = P((Axb + lmb[i]).T).T
            prim = Axb - z_new
            dual = rho * A.T @ (z_new - z[i])
            lmb[i] = lmb[i] + prim
            z[i] = z_new
            p_max = max(p_max, float(np.max(np.linalg.norm(prim, axis=0))))
            d_max = max(d_max, float(np.max(np.linalg.norm(dual, axis=0))))
        state["x"] = xk
        return p_max, d_max

    _consensus_residual_loop(update, max_iter, threshold)
    out = state["x"].T
    return out[0] if single else out


def project_soc(z0, A, b, rho=1e0, max_iter=100, tol=1e-5, verbose=False):
    """Projection onto {z : A z + b in SOC} by ADMM (isls/projections.py:163-234)."""
    single = z0.ndim == 1
    Z0 = (z0[None] if single else z0).T
    z = Z0.copy()
    lmb = np.zeros((A.shape[0], Z0.shape[1]))
    lhs_inv = np.linalg.inv(np.eye(Z0.shape[0]) + rho * A.T @ A)
    state = {"z": z}

    def update():
        zk = state["z"]
        x = project_soc_unit((A @ zk + b[:, None] + lmb).T).T
        z_new = lhs_inv @ (Z0 + rho * A.T @ (-b[:, None] + x - lmb))
        prim = A @ z_new + b[:, None] - x
        dual = rho * (z_new - zk)
        lmb[...] = lmb + prim
        state["z"] = z_new
        return float(np.max(np.linalg.norm(prim, axis=0))), float(np.max(np.linalg.norm(dual, axis=0)))

    _consensus_residual_loop(update, max_iter, tol)
    out = state["z"].T
    return out[0] if single else out


def project_set_convex_dykstra(x0, projections=[], max_iter=200, tol=1e-4, verbose=False):
    """Dykstra's alternating projections (isls/projections.py:465-504)."""
    single = x0.ndim == 1
    u = (x0[None] if single else x0).copy()
    z = np.zeros((len(projections),) + u.shape)
    k, cI = 0, np.full(u.shape[0], 10.0)
    while k <= max_iter and np.any(cI >= tol):
        cI = np.zeros(u.shape[0])
        for i, P in enumerate(projections):
            prev_u, prev_z = u, z[i].copy()
            u = P(prev_u - prev_z)
            z[i] = u - (prev_u - prev_z)
            cI += np.linalg.norm(prev_z - z[i], axis=-1) ** 2
        k += 1
    return u
