"""Slow path for arbitrary user callbacks (SURVEY 7 "torch-callable slow path"): `forward_model(x, u)`, `cost_function(x, u)`
and `get_Cs(x, u)` given as plain Python callables in the reference's conventions.

A HIP kernel cannot call back into Python, so when the forward model or the cost is a callable the LINE SEARCH leaves the
device: the Riccati gain / feed-forward passes and the ADMM update stay on the GPU, the candidates are rolled out on the
host through the callable -- the reference's `rollout_DP` (isls/isls.py:310-334), vectorised over all B x L candidate rows
at once -- their costs come from the callable (or the via-point cost, isls/sls_base.py:25-44) plus the augmented-Lagrangian
terms of the ADMM line search (isls.py:471-476), and arg-min / NaN rule / acceptance test follow isls.py:357-369.  One
host round trip per line search: the notebooks run unchanged, at host speed.

Callables must act row-wise on stacked rows (`f(x[R, n], u[R, m]) -> [R, n]`), which is how the reference calls them with
its L candidates (every model of the notebooks does).
"""
import numpy as np
import torch

from . import _capi as capi


def via_point_cost(sls, x, u, b):
    """SLSBase.compute_cost for the candidates of trajectory b: x [L,N,n], u [L,N,m] -> [L] (no 1/2)."""
    zs = sls.zs if sls.zs.ndim == 2 else sls.zs[b]
    dx = x - zs[sls.seq]
    c = np.einsum("lti,tij,ltj->l", dx, sls.Qs[sls.seq], dx)
    return c + sls.u_std * np.sum(u * u, axis=(-1, -2))


def candidate_costs(sls, x, u, b):
    """The reference's `self.cost_function(x_noms, u_noms)` for trajectory b."""
    fn = sls._cost_function
    if fn is None:
        return via_point_cost(sls, x, u, b)
    return np.asarray(fn(x, u), dtype=np.float64).reshape(-1)


def nominal_cost(sls):
    """cost of the current nominal of every trajectory through the user's cost function (nominal_values setter)."""
    e = sls.engine
    xs, us = e.xhat.cpu().numpy().astype(np.float64), e.uhat.cpu().numpy().astype(np.float64)
    return np.array([float(np.asarray(candidate_costs(sls, xs[b][None], us[b][None], b)).reshape(-1)[0]) for b in range(sls.batch)])


def line_search(sls, L, flags=0, active=None, K=None, k=None, plain_only=False):
    """Host line search over alphas[:L]; leaves the winner (or the kept nominal) in engine.xx / xu, its plain cost in
    cost_new, the index in best and the status bits, exactly like isls_rollout_ls_* does.  K / k override the engine's gains
    (isls_admm searches open loop along d_u: zero gains); plain_only drops the augmented-Lagrangian terms (isls.py:593-606)."""
    e = sls.engine
    B, N, n, m = sls.batch, sls.N, sls.x_dim, sls.u_dim
    f = sls._forward_model
    host = lambda t: t.detach().cpu().numpy().astype(np.float64)              # noqa: E731
    xh, uh = host(e.xhat), host(e.uhat)
    K = host(e.K) if K is None else np.asarray(K, dtype=np.float64)
    k = host(e.k) if k is None else np.asarray(k, dtype=np.float64)
    act = np.ones(B, dtype=bool) if active is None else active.cpu().numpy().astype(bool)
    alphas = np.asarray(sls.alphas[:L], dtype=np.float64)
    x = np.repeat(xh[:, None, 0, :], L, axis=1)                                # [B, L, n]
    x_log, u_log = np.zeros((B, L, N, n)), np.zeros((B, L, N, m))
    for t in range(N):
        dx = x - xh[:, None, t, :]
        u = (np.einsum("blj,bij->bli", dx, K[:, t]) + alphas[None, :, None] * k[:, None, t, :]) + uh[:, None, t, :]
        x_log[:, :, t], u_log[:, :, t] = x, u
        x = np.asarray(f(x.reshape(B * L, n), u.reshape(B * L, m)), dtype=np.float64).reshape(B, L, n)
    wq = None if e.wq is None else np.broadcast_to(host(e.wq), (B, N, n)) if e.wq.ndim == 3 else np.broadcast_to(host(e.wq)[None], (B, N, n))
    wr = None if e.wr is None else np.broadcast_to(host(e.wr), (B, N, m)) if e.wr.ndim == 3 else np.broadcast_to(host(e.wr)[None], (B, N, m))
    rx = host(e.zx) - host(e.lx) if e.zx is not None and wq is not None and not plain_only else None
    ru = host(e.zu) - host(e.lu) if e.zu is not None and wr is not None and not plain_only else None
    cost_cur = host(e.cost)
    xx, xu = host(e.xx), host(e.xu)
    best, cost_new, status = np.zeros(B, dtype=np.int32), host(e.cost_new), np.zeros(B, dtype=np.int32)
    for b in range(B):
        if not act[b]:
            continue
        plain = candidate_costs(sls, x_log[b], u_log[b], b).astype(np.float64).copy()
        aug = plain.copy()
        if rx is not None:                                                      # (dx*dx) @ Qr: row sums of the weights
            aug += np.sum((x_log[b] - rx[b][None]) ** 2 * wq[b][None], axis=(-1, -2))
        if ru is not None:
            aug += np.sum((u_log[b] - ru[b][None]) ** 2 * wr[b][None], axis=(-1, -2))
        nan = np.isnan(aug)
        if nan.any():
            status[b] |= capi.ST_NAN_COST
            if flags & capi.RO_NAN_TO_1E5:
                aug[nan], plain[nan] = 1e5, 1e5
        ind = int(np.argmin(aug))
        accept = True
        if flags & capi.RO_ACCEPT_TEST:
            accept = (plain[ind] - cost_cur[b]) < 0.0
            if not accept:
                status[b] |= capi.ST_LS_REJECT
        best[b] = ind
        cost_new[b] = plain[ind] if accept else cost_cur[b]
        xx[b], xu[b] = (x_log[b, ind], u_log[b, ind]) if accept else (xh[b], uh[b])
    e.xx.copy_(e._t(xx)), e.xu.copy_(e._t(xu)), e.cost_new.copy_(e._t(cost_new))
    e.best.copy_(torch.as_tensor(best, device=e.device))
    e.status.bitwise_or_(torch.as_tensor(status, device=e.device))


def expansion(sls, get_Cs):
    """(Cts [B,N,n+m,n+m], cts [B,N,n+m]) of the user's cost about the current nominal: `cts, Cts = get_Cs(x_nom, u_nom)`
    per trajectory (isls/isls.py:102)."""
    e = sls.engine
    xs, us = e.xhat.cpu().numpy().astype(np.float64), e.uhat.cpu().numpy().astype(np.float64)
    out = [get_Cs(xs[b], us[b]) for b in range(sls.batch)]
    return np.stack([np.asarray(o[1], dtype=np.float64) for o in out]), np.stack([np.asarray(o[0], dtype=np.float64) for o in out])


# ---- closed loops with process noise: host loops in the reference's own order of random draws --------------------------------
def noisy_closed_loop(f, x0, N, u_dim, control, noise_scale):
    """The loops of isls/isls_base.py:28-71 and isls/sls_base.py:61-105 with process noise: `w = np.random.normal(0,
    noise_scale, x0.shape)` is drawn from numpy's global generator once per step, in the reference's order, so a seeded run
    reproduces the reference's trajectories.  `control(i, x_log) -> u_i [M, m]`.  Returns (x_log [M,N,n], u_log [M,N,m])."""
    x0 = np.asarray(x0, dtype=np.float64)
    single = x0.ndim == 1
    xs = np.atleast_2d(x0)
    M, n = xs.shape
    x_log, u_log = np.zeros((M, N + 1, n)), np.zeros((M, N, u_dim))
    x_log[:, 0] = xs
    for i in range(N):
        u_log[:, i] = control(i, x_log)
        w = np.random.normal(loc=0, scale=noise_scale, size=x0.shape)
        x_log[:, i + 1] = np.asarray(f(x_log[:, i], u_log[:, i]), dtype=np.float64) + w
    return (x_log[0, :-1], u_log[0]) if single else (x_log[:, :-1], u_log)
