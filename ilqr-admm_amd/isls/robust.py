"""Batched `iSLS.isls_admm` (isls/isls.py:503-712): iterative SLS with feedback columns [d, phi] and ADMM on their rows.

The reference solves, per outer iteration, one dense (N m)^2 system for the 1 + dim columns and multiplies by the dense
transfer matrices.  Here every column is the minimiser of a time-varying LQ problem about the nominal (see
`isls_columns_args` in include/isls_hip.h), so the x-step of an ADMM iteration is

    C feed-forward passes (isls_riccati_ff)  ->  isls_columns_rollout  ->  open-loop line search on column 0 (isls_rollout_ls
    with zero gains: `rollout_batch(x_nom, u_nom + alpha d_u)`, isls.py:593-606)

after ONE Riccati gain pass per outer iteration, and the z-step is isls_columns_admm around the row projection
(isls_project_rows for `projections.ConvexSets`, the caller's numpy function otherwise).  Results equal the dense form up to
rounding, including the last control (SURVEY 8a quirk i), which the dense form sets from its own cost term.
"""
import numpy as np
import torch

from . import _capi as capi
from .engine import _stream_ptr
from .projections import ConvexSets


def _row_projection(project, d, C):
    """project_x / project_u of isls_admm -> None, ConvexSets on rows of dimension C (device) or a callable (host)."""
    if project is False or project is None:
        return None
    if isinstance(project, ConvexSets):
        if project.dim != C or project.cols != (0, C):
            raise ValueError(f"isls_admm projects rows [d, phi] of dimension {C}; the ConvexSets acts on "
                             f"columns {project.cols} of rows of dimension {project.dim}")
        return project
    if callable(project):
        return project
    raise TypeError("project_x / project_u must be False, a projections.ConvexSets or a callable (rows, nominal) -> rows")


def isls_admm(self, dim, get_AB=None, get_Cs=None, project_x=False, project_u=False, max_admm_iter=20, k_max=20,
              max_line_search=20, rho_x=None, rho_u=None, alpha=1, threshold=1e-3, verbose=False, log=False):
    """Returns (du [N m], phi_u [N m, dim]) of the last ADMM x-step (with a leading batch axis when batch > 1).

    project_x / project_u: `projections.ConvexSets` over the rows [nominal + d, phi] (the shift by the nominal of
    notebook cell 25 is applied on the device), or a callable `(rows [N d, 1 + dim], nominal [N, d]) -> rows` in the
    reference's convention (host round trip per ADMM iteration).  `self.admm_iters` holds the executed ADMM iterations of
    the last outer iteration per problem, `self.admm_logs` their (prim, dual) residuals [J, B, 2]."""
    self._check_get_Cs(get_Cs)
    e = self.engine
    B, N, n, m, C = self.batch, self.N, self.x_dim, self.u_dim, int(dim) + 1
    if not 1 <= dim <= n or C > capi.MAX_ROW_DIM:
        raise ValueError(f"dim must be in [1, {min(n, capi.MAX_ROW_DIM - 1)}]")
    px, pu = _row_projection(project_x, n, C), _row_projection(project_u, m, C)
    free = (-np.inf, np.inf)
    e.set_admm(rho_x=rho_x if px is not None else None, rho_u=rho_u if pu is not None else None,
               x_box=free if px is not None else None, u_box=free if pu is not None else None, relax=alpha)
    kern, sfx = e.kern, e.sfx
    z = lambda *s: torch.zeros(*s, dtype=e.dtype, device=e.device)            # noqa: E731
    kcol, dx, du = z(C, B, N, m), z(C, B, N, n), z(C, B, N, m)
    zero_x, zero_u, zero_K = z(1, 1, n), z(1, 1, m), z(B, N, m, n)
    blocks = {}
    for key, proj, d, W, nom in (("x", px, n, e.Qr, e.xhat), ("u", pu, m, e.Rr, e.uhat)):
        if proj is None:
            blocks[key] = None
            continue
        blk = dict(xx=dx if key == "x" else du, z=z(C, B, N, d), l=z(C, B, N, d), z_prev=z(C, B, N, d),
                   work=z(B, N * d, C), W=W, nom=nom if isinstance(proj, ConvexSets) else None, proj=proj, desc=None)
        if isinstance(proj, ConvexSets):
            sets = [{k_: (e._t(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k_, v in st.items()}
                    for st in proj.sets]
            blk["desc"] = capi.Kernels.project_args(blk["work"], blk["work"], sets, rho=proj.rho, max_iter=proj.max_iter,
                                                    threshold=proj.threshold, active=e.admm_active)
        blocks[key] = blk
    bx, bu = blocks["x"], blocks["u"]
    constrained = bx is not None or bu is not None
    J = int(max_admm_iter) if constrained else 1
    L = int(max_line_search)
    logbuf = z(J, B, 2)
    e.outer_active.fill_(1)
    hist = [[float(c)] for c in np.atleast_1d(np.asarray(self.cost, dtype=np.float64))]
    mask3 = lambda a: a.to(torch.bool).view(B, 1, 1)                          # noqa: E731
    for k in range(k_max):
        self._linearize(get_AB)
        e.expand()
        rec = e.ff_record()                                                     # packed step records for the C x J ff passes
        e.gain(active=e.outer_active, rec=rec)
        seg = e.ff_seg()                                                        # time-parallel feed-forward passes (isls_ffseg)
        if seg is not None:
            e.feedforward_prepare(seg, active=e.outer_active)
        Cuu = e.hessians()[1]
        e.admm_active.copy_(e.outer_active)
        e.admm_iters.zero_()
        e.res_prev.fill_(1e6)
        for blk in (bx, bu):
            if blk is not None:
                blk["l"].zero_()                                                # lmb restarts, z is warm-started (isls.py:613-616)
        for j in range(J):
            act = e.admm_active
            for c in range(C):                                                  # STEP 1: the columns' feed-forward terms
                kern.riccati_ff(e.A, e.Bm, e.c0x if c == 0 else zero_x, e.c0u if c == 0 else zero_u, e.K, e.Quu, e.fac,
                                e.Qux, kcol[c], Qr=e.Qr if bx is not None else None, Rr=e.Rr if bu is not None else None,
                                zx=bx["z"][c] if bx is not None else None, lx=bx["l"][c] if bx is not None else None,
                                zu=bu["z"][c] if bu is not None else None, lu=bu["l"][c] if bu is not None else None,
                                solve_mode=e.solve_mode, active=act, seg=seg, rec=rec, stream=_stream_ptr())
            kern.columns_rollout(e.A, e.Bm, Cuu, e.c0u, e.K, kcol, dx, du, Rr=e.Rr if bu is not None else None,
                                 zu=bu["z"] if bu is not None else None, lu=bu["l"] if bu is not None else None,
                                 active=act, stream=_stream_ptr())
            # line search on d_u: open-loop rollouts of u_nom + alpha d_u, plain cost, first arg-min (isls.py:593-606)
            kern.rollout_ls(e.model, e.model_par, zero_K, du[0], e.xhat, e.uhat, e.alphas[:L], e.Qtab, e.ztab, e.seq, e.u_std,
                            e.xx, e.xu, best=e.best, cost_new=e.cost_new, flags=0, status=e.status, active=act,
                            q_nonzero=e.q_nonzero, cost_model=e.cost_model, cost_par=e.cost_par, stream=_stream_ptr())
            on = mask3(act)
            step = torch.where(act.to(torch.bool), e.alphas[:L][e.best.long()], torch.ones_like(e.cost_new))
            du[0].mul_(step.view(B, 1, 1))                                      # du_opt[:, 0] = alpha* d_u
            dx[0].copy_(torch.where(on, e.xx - e.xhat, dx[0]))                  # dx_opt[:, 0] = x_noms[ind] - x_nom
            if not constrained:
                e.admm_iters.add_(act)
                break
            # STEP 2: z = Proj(alpha x + (1 - alpha) z + lmb), lmb += x - z, residuals and stop rules (isls.py:626-665)
            dims = (B, N, n, m, C)
            kern.columns_admm(0, dims, e.res, e.res_prev, x=bx, u=bu, relax=alpha, active=act, stream=_stream_ptr())
            for blk, nom in ((bx, e.xhat), (bu, e.uhat)):
                if blk is None:
                    continue
                if blk["desc"] is not None:
                    kern._call("project_rows", sfx, blk["desc"], _stream_ptr())
                else:                                                           # the caller's numpy projection, problem by problem
                    rows, nom_h, on_h = blk["work"].cpu().numpy(), nom.cpu().numpy(), act.cpu().numpy()
                    for b in range(B):
                        if on_h[b]:
                            rows[b] = np.asarray(blk["proj"](rows[b].copy(), nom_h[b]), dtype=np.float64)
                    blk["work"].copy_(e._t(rows))
            kern.columns_admm(1, dims, e.res, e.res_prev, x=bx, u=bu, relax=alpha, tol_abs=threshold, tol_rel=1e-3,
                              active=act, iters=e.admm_iters, stream=_stream_ptr())
            logbuf[j].copy_(e.res)
            if not bool(act.any().item()):
                break
        # new nominal: x_nom + d_x, u_nom + d_u of the last x-step (isls.py:684-687); the setter evaluates its cost
        oa = mask3(e.outer_active)
        e.xhat.copy_(torch.where(oa, e.xhat + dx[0], e.xhat))
        e.uhat.copy_(torch.where(oa, e.uhat + du[0], e.uhat))
        e.evaluate_cost()
        st = e.status.cpu().numpy()
        if (st & capi.ST_NOT_PD).any():
            raise np.linalg.LinAlgError("Quu not positive definite")
        cost = e.cost.cpu().numpy().astype(np.float64)
        self.cost_log.append(self.cost)
        active = e.outer_active.cpu().numpy().astype(bool)
        for b in np.nonzero(active)[0]:
            prev = hist[b][-1]
            hist[b].append(float(cost[b]))
            if verbose:
                print("Iteration number ", k, "iSLS cost: ", cost[b])
            stop = abs(cost[b] - prev) < 1e-4                                   # isls.py:695-697
            if not stop and len(hist[b]) >= 5:                                  # oscillation test, isls.py:699-701 (NaN on short logs)
                stop = abs(np.mean(hist[b][-4:]) - np.mean(hist[b][-8:-4])) < 1e-3
            active[b] = not stop
        e.outer_active.copy_(torch.as_tensor(active.astype(np.int32), device=e.device))
        if not active.any():
            break
    self.admm_iters = e.admm_iters.cpu().numpy()
    self.admm_logs = logbuf.cpu().numpy()
    self.outer_iters = np.array([len(h) - 1 for h in hist])
    du_out = du[0].reshape(B, N * m).cpu().numpy()
    phi_out = du[1:].permute(1, 2, 3, 0).reshape(B, N * m, C - 1).cpu().numpy()
    self._dx_columns = dx.permute(1, 2, 3, 0).reshape(B, N * n, C).cpu().numpy()
    return (du_out[0], phi_out[0]) if B == 1 else (du_out, phi_out)
