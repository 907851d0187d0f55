#!/usr/bin/env python3
"""Config 5 of BASELINE.json / SURVEY 8(d): SLS-ADMM with SOC chance constraints, N=50, B=8192 problems that differ in
target, bound and variance, fp32 (and fp64).  Prints one JSON line per precision: ADMM iterations/s over the whole batch
(max_iter fixed, stop rules disabled so the work is constant), with the CPU oracle timed beside it on a bounded sample.

    python tools/bench_config5.py [--batch 8192] [--dim 1|3] [--iters 50]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ilqr-admm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch


def problem(B, nb_dim, N, seed=0):
    from isls import sls_dense as dense
    from isls.utils import get_double_integrator_AB
    pj = sys.modules["isls.projections"]
    rng = np.random.default_rng(seed)
    n, m, p = 2 * nb_dim, nb_dim, nb_dim
    A, Bm = get_double_integrator_AB(nb_dim, nb_deriv=2, dt=1.0 / N)
    Sw, Su = dense.transfer_matrices(A, Bm, N)
    targets = np.concatenate([rng.uniform(0.5, 1.5, (B, nb_dim)), np.zeros((B, nb_dim))], 1)
    zs = np.stack([np.zeros((B, n)), targets], 1)
    seq = np.zeros(N, dtype=np.int32); seq[N - 1] = 1
    Q, R, xd = dense.dense_cost(zs, np.stack([np.zeros((n, n)), 1e6 * np.eye(n)]), seq, 1e-2, N, n, m)
    rr = dense.rho_diagonal(1e2, N, m)
    Linv, r_side = dense.admm_sls_setup(Sw, Su, Q, R, xd, rr, p, B)
    cs = pj.chance_constraint_rows(p, rng.uniform(5.0, 8.0, B), -rng.uniform(5.0, 8.0, B), rng.uniform(0.005, 0.02, B),
                                   1.6448536269514722)
    return Linv, r_side, rr, cs


def run(kern, Linv, r_side, rr, cs, iters, dtype, wrap, sync):
    mk = lambda a: wrap(np.ascontiguousarray(a, dtype=dtype))   # noqa: E731
    sets = [{k: (mk(v) if isinstance(v, np.ndarray) else v) for k, v in st.items()} for st in cs.sets]
    x_u = wrap(np.zeros(r_side.shape, dtype=dtype))
    it = wrap(np.zeros(r_side.shape[0], dtype=np.int32))
    args = (mk(Linv), mk(r_side), mk(rr), sets, x_u)
    kw = dict(alpha=1.0, tol=0.0, rel_tol=0.0, max_iter=iters, rho=cs.rho, inner_max_iter=cs.max_iter, threshold=cs.threshold, iters=it)
    kern.sls_admm(*args, **kw)
    sync()
    t0 = time.perf_counter()
    reps = 0
    while True:
        kern.sls_admm(*args, **kw)
        reps += 1
        sync()
        if time.perf_counter() - t0 > 2.0 or reps >= 20:
            break
    return (time.perf_counter() - t0) / reps, x_u


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--dim", type=int, default=1)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--horizon", type=int, default=50)
    args = ap.parse_args()
    import isls  # noqa: F401
    from isls.engine import kernels
    from oracle import oracle as orc
    B, N = args.batch, args.horizon
    Linv, r_side, rr, cs = problem(B, args.dim, N)
    hip = kernels()
    dev = lambda a: torch.from_numpy(a).cuda()                  # noqa: E731
    okern, olib = orc.load()
    cores = orc.set_threads(olib, min(16, len(os.sched_getaffinity(0))))
    sample = min(B, 1024)
    cs_s = type(cs)(cs.dim, cs.cols, [{k: (v[:sample] if isinstance(v, np.ndarray) and v.ndim == 3 or isinstance(v, np.ndarray) and k == "b" else v)
                                       for k, v in st.items()} for st in cs.sets], rho=cs.rho, max_iter=cs.max_iter, threshold=cs.threshold)
    for dtype, name in ((np.float32, "f32"), (np.float64, "f64")):
        dt, x_u = run(hip, Linv, r_side, rr, cs, args.iters, dtype, dev, torch.cuda.synchronize)
        dtc, x_c = run(okern, Linv, r_side[:sample], rr, cs_s, args.iters, dtype, lambda a: a, lambda: None)
        d = np.abs(x_u.cpu().numpy()[:sample].astype(np.float64) - x_c.astype(np.float64)).reshape(sample, -1).max(1)
        d = d / np.abs(x_c.astype(np.float64)).reshape(sample, -1).max(1)
        # problems whose bound is infeasible do not contract and amplify rounding differences: report median and max
        err = {"median": float(np.median(d)), "max": float(np.max(d))}
        print(json.dumps({"metric": "SLS-ADMM iterations/sec (config 5)", "value": args.iters / dt, "unit": "iterations/s",
                          "dtype": name, "config": {"workload": f"config5: DI-{args.dim}D SLS-ADMM, SOC chance constraints",
                                                    "batch": B, "horizon": N, "admm_iters": args.iters, "inner_max_iter": cs.max_iter},
                          "problems_per_s": B / dt, "ms_per_solve": 1e3 * dt,
                          "cpu_baseline": {"value": args.iters / (dtc * B / sample), "unit": "iterations/s", "cores": cores, "kind": "port",
                                           "sample": f"{sample} problems, scaled linearly to {B}"},
                          "rel_diff_vs_oracle_on_sample": err}))


if __name__ == "__main__":
    main()
