#!/usr/bin/env python3
"""Experiment (not part of the product): the headline batch split into `parts` sub-batches that run their outer
iterations concurrently on separate HIP streams, against the single-stream run of the whole batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ilqr-admm_amd"), ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import isls_problems as P
from isls import models
from isls.engine import Engine


def make(B, N, J, L, seed, dev):
    cfg = P.config2(batch=B, N=N, seed=seed)
    n, m = cfg["n"], cfg["m"]
    eng = Engine(B, N, n, m, dtype=torch.float64, device=dev)
    mdl = models.LTI(cfg["A"], cfg["B"])
    eng.set_model(mdl.model_id, mdl.params())
    eng.set_quadratic_cost(cfg["zs"], cfg["Qs"], cfg["seq"], cfg["u_std"])
    eng.set_nominal(np.repeat(cfg["x0"][:, None, :], N, axis=1), cfg["u0"])
    eng.set_admm(rho_u=cfg["rho_u"], u_box=(cfg["u_lo"], cfg["u_hi"]), relax=cfg["relax"])
    eng.build_outer(L, J, tol_abs=0.0, tol_rel=0.0)
    return eng


def step(eng):
    eng.linearize(); eng.expand(); eng.run_outer(); eng.accept_x_step(); eng.reduce()


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    B, N, J, L, K = 4096, 100, 5, 20, 10
    for parts in (1, 2, 4):
        engs = [make(B // parts, N, J, L, i, dev) for i in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        for _ in range(2):
            for e, s in zip(engs, streams):
                with torch.cuda.stream(s):
                    step(e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            for e, s in zip(engs, streams):
                with torch.cuda.stream(s):
                    step(e)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"parts {parts}: {1e3 * dt / K:.3f} ms per outer iteration of {B} trajectories  ({K / dt:.1f} it/s)", flush=True)
        del engs


if __name__ == "__main__":
    main()
