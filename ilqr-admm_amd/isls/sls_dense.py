"""Dense (N m)^2 set-up of system level synthesis (config 5) -- host numpy, no device state.

Everything here is cold set-up that the reference also does once per problem class with dense numpy (isls/base.py:29-50,
98-119; isls/sls.py:205-242, 325-352): transfer matrices, the unconstrained SLS solution, the inverse used by every ADMM
iteration and the controller K = Phi_u Phi_x^-1.  The ADMM iterations themselves run on the device (isls_sls_admm).
"""
import numpy as np


def transfer_matrices(A, B, N):
    """Sw [N n, N n] and Su [N n, N m] with x = Sw[:, :n] x0 + Su u (isls/base.py:20-21, 98-119): block (i, j) of Sw is
    A^(i-j) for i >= j and block (i, j) of Su is A^(i-j-1) B for i > j, the powers formed by right-multiplication like the
    reference's column-by-column recursion."""
    n, m = A.shape[0], B.shape[1]
    pw = [np.eye(n)]
    for _ in range(N - 1):
        pw.append(pw[-1] @ A)
    Sw, Su = np.zeros((N * n, N * n)), np.zeros((N * n, N * m))
    for i in range(N):
        for j in range(i + 1):
            Sw[i * n:(i + 1) * n, j * n:(j + 1) * n] = pw[i - j]
            if j < i:
                Su[i * n:(i + 1) * n, j * m:(j + 1) * m] = pw[i - j - 1] @ B
    return Sw, Su


def transfer_matrices_ltv(A, B):
    """Sw, Su of a time-varying linearisation A [N,n,n], B [N,n,m] (the `AB` setter, isls/base.py:98-119): block (i, j) of
    Sw is A_{i-1} ... A_j, block (i, j) of Su is A_{i-1} ... A_{j+1} B_j, built row block by row block."""
    N, n, m = A.shape[0], A.shape[1], B.shape[2]
    Sw, Su = np.zeros((N * n, N * n)), np.zeros((N * n, N * m))
    Sw[:n, :n] = np.eye(n)
    for i in range(1, N):
        r, pr = slice(i * n, (i + 1) * n), slice((i - 1) * n, i * n)
        Sw[r, :i * n] = A[i - 1] @ Sw[pr, :i * n]
        Sw[r, r] = np.eye(n)
        Su[r, :i * m] = A[i - 1] @ Su[pr, :i * m]
        Su[r, (i - 1) * m:i * m] = B[i - 1]
    return Sw, Su


def dense_cost(zs, Qs, seq, u_std, N, n, m):
    """Q [N n, N n] block diagonal, R [N m, N m] = u_std I, stacked targets xd [..., N n] (isls/base.py:81-89)."""
    Q = np.zeros((N * n, N * n))
    for t in range(N):
        Q[t * n:(t + 1) * n, t * n:(t + 1) * n] = Qs[seq[t]]
    xd = zs[..., seq, :].reshape(zs.shape[:-2] + (N * n,))
    return Q, u_std * np.eye(N * m), xd


def compute_inverses(M, m, N):
    """[M^-1, M[m:, m:]^-1, M[2m:, 2m:]^-1, ...] by successive rank-2m down-dates of the first inverse
    (isls/base.py:29-50): striking the first m rows and columns of M is the update M - U V with the 2m-column U, V below,
    so the Woodbury identity gives the trailing inverse from the previous one."""
    def strike(Mi, Mi_inv):
        d = Mi.shape[0]
        U, V = np.zeros((d, 2 * m)), np.zeros((2 * m, d))
        U[:m, :m] = np.eye(m)
        U[m:, m:] = Mi[m:, :m]
        V[:m, m:] = Mi[:m, m:]
        V[m:, :m] = np.eye(m)
        core = np.linalg.inv(np.eye(2 * m) - V @ Mi_inv @ U)
        return (Mi_inv + Mi_inv @ U @ core @ V @ Mi_inv)[m:, m:]
    out = [np.linalg.inv(M)]
    for i in range(N):
        out.append(strike(M[i * m:, i * m:], out[i]))
    return out


def solve_sls(Sw, Su, Q, R, xd, N, n, m, l_side_invs=None):
    """Unconstrained SLS (isls/sls.py:205-233): du [..., N m], block-lower-triangular PHI_U [N m, N n], the inverses."""
    DTQ = Su.T @ Q
    if l_side_invs is None:
        l_side_invs = compute_inverses(DTQ @ Su + R, m, N)
    du = (l_side_invs[0] @ DTQ @ xd[..., None])[..., 0]
    r_side = -DTQ @ Sw
    PHI_U = np.zeros((N * m, N * n))
    for i in range(N):
        PHI_U[i * m:, i * n:(i + 1) * n] = l_side_invs[i] @ r_side[i * m:, i * n:(i + 1) * n]
    return PHI_U, du, l_side_invs


def controller(Sw, Su, PHI_U, du):
    """K = Phi_u Phi_x^-1, k = (I - K Su) du (isls/sls.py:235-242)."""
    K = PHI_U @ np.linalg.inv(Sw + Su @ PHI_U)
    return K, (np.eye(Su.shape[1]) - K @ Su) @ du


def rho_diagonal(rho_u, N, m):
    """Diagonal of the reference's block-diagonal Rr (isls/base.py:55-79, dp=False) as a vector [N m]."""
    if isinstance(rho_u, (int, float)):
        return np.full(N * m, float(rho_u))
    r = np.asarray(rho_u, dtype=np.float64)
    if r.ndim == 2 and r.shape == (m, m) and np.count_nonzero(r - np.diag(np.diag(r))) == 0:
        return np.tile(np.diag(r), N)
    if r.ndim == 3 and all(np.count_nonzero(b - np.diag(np.diag(b))) == 0 for b in r):
        return np.concatenate([np.diag(b) for b in r])
    raise NotImplementedError("ADMM_SLS on the device takes diagonal rho_u weights")


def admm_sls_setup(Sw, Su, Q, R, xd, rr, p, batch):
    """Operands of the ADMM_SLS iteration (isls/sls.py:329-352, 371): the shared inverse (Su'Q Su + R + Rr)^-1 [N m, N m]
    and the right-hand sides r_side [B, N m, 1 + p] = [Su'Q xd_b, -Su'Q Sx] with Sx the x0-position columns of Sw."""
    SuTQ = Su.T @ Q
    l_side_inv = np.linalg.inv(SuTQ @ Su + R + np.diag(rr))
    r_ff = (SuTQ @ np.broadcast_to(xd, (batch, xd.shape[-1]))[..., None])[..., 0]
    r_fb = -SuTQ @ Sw[:, :p]
    r_side = np.concatenate([r_ff[..., None], np.broadcast_to(r_fb, (batch,) + r_fb.shape)], axis=-1)
    return l_side_inv, np.ascontiguousarray(r_side)
