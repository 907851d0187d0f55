"""Generic two-block ADMM driver with callbacks, the host-side mirror of the reference's `ADMM()`
(isls/admm.py:6-106): same arguments, same return tuple, same two stop rules, same in-place update of the
caller's `lmb_*_init` arrays.

On the hot path this loop does NOT run: `iSLS.ilqr_admm` / `SLS.ADMM_LQT_DP` with box descriptors execute the
z-step, dual step, residual norms and stop rules inside the HIP kernel `admm_update`.  This function is the
generic route for projections that have no device kernel yet (arbitrary Python callables): the x-step callback
still runs the HIP Riccati/rollout kernels, the projection runs wherever the caller's function runs.
"""
import numpy as np


def ADMM(shape_x, shape_u, f_argmin, project_x=False, project_u=False, z_x_init=None, z_u_init=None,
         lmb_x_init=None, lmb_u_init=None, Qr=None, Rr=None, max_iter=20, alpha=1., tol=1e-3, verbose=False,
         return_lmb=False, log=False):
    logs = []
    blocks = {}
    for name, proj, shape, z0, l0 in (("x", project_x, shape_x, z_x_init, lmb_x_init),
                                      ("u", project_u, shape_u, z_u_init, lmb_u_init)):
        if proj:
            blocks[name] = dict(proj=proj, z=np.zeros(shape) if z0 is None else z0,
                                lmb=np.zeros(shape) if l0 is None else l0)
    prim, dual = 1e6, 1e6
    ret = ()
    for j in range(max_iter):
        reg = {n: blk["z"] - blk["lmb"] for n, blk in blocks.items()}
        ret = f_argmin(reg.get("x"), reg.get("u"))
        if not ret:
            ret = (ret,)
            print("unsuccesful first step of ADMM at iteration", j)
            break
        step = {"x": ret[0], "u": ret[1]}
        prev_prim, prev_dual = prim, dual
        prim = dual = 0.
        for n, blk in blocks.items():
            z_prev = blk["z"]
            blk["z"] = blk["proj"](alpha * step[n] + (1 - alpha) * z_prev + blk["lmb"])
            r = step[n] - blk["z"]
            blk["lmb"] += r                                   # in place, like the reference (admm.py:52,59)
            prim += np.linalg.norm(r)
            dual += np.linalg.norm(blk["z"] - z_prev)
        logs.append(np.array([prim, dual]))
        if prim < tol and dual < tol:
            if verbose:
                print("ADMM converged at iteration ", j, "!")
                print("ADMM residual is ", "{:.2e}".format(prim), "{:.2e}".format(dual))
            break
        if abs(prev_prim - prim) / (prev_prim + 1e-30) < tol and abs(prev_dual - dual) / (prev_dual + 1e-30) < tol:
            if verbose:
                print("ADMM can't improve anymore at iteration ", j, "!")
                print("ADMM residual is ", "{:.2e}".format(prim), "{:.2e}".format(dual))
            break
        if j == max_iter - 1 and verbose:
            print("ADMM residuals-> primal:", "{:.2e}".format(prim), "dual:", "{:.2e}".format(dual))
            print("ADMM: Max iteration reached.")
    out = tuple(ret)
    if return_lmb:
        bx, bu = blocks.get("x"), blocks.get("u")
        out += (bx["lmb"] if bx else None, bu["lmb"] if bu else None, bx["z"] if bx else None, bu["z"] if bu else None)
    if log:
        out += (logs,)
    return out
