#!/usr/bin/env python3
"""Outer iterations per second of BASELINE.json's configs 3 (3R arm, B = 4096, N = 100, state + control boxes, 10 ADMM
iterations x 5 candidates) and 4 (car, B = 4096 per GPU, N = 200, control box + two keep-out rectangles, 5 x 20) through the
class surface (`iSLS.ilqr_admm`, tol = 0: fixed work):
    python tools/config_bench.py [--config 3|4] [--outer 6]
Under `rocprofv3 --kernel-trace --stats` the same run gives the kernel split."""
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

import isls_problems as P
from test_full_size import _make


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--outer", type=int, default=6)
    ap.add_argument("--batch", type=int, default=4096)
    a = ap.parse_args()
    from isls import Box
    pj = sys.modules["isls.projections"]
    B = a.batch
    if a.config == 3:
        cfg = P.config3(batch=B, N=100, seed=0)
        kw = dict(project_x=Box(cfg["x_lo"], cfg["x_hi"]), project_u=Box(cfg["u_lo"], cfg["u_hi"]),
                  max_line_search_iter=cfg["max_line_search"], max_admm_iter=cfg["max_admm_iter"], rho_x=cfg["rho_x"],
                  rho_u=cfg["rho_u"], alpha=1.0, tol=0.0)
    else:
        cfg = P.config4(batch=B, N=200, seed=0)
        rho_x = np.zeros((200, 4, 4)); rho_x[:, :2, :2] = 0.1 * np.eye(2)
        cs = pj.keepout_rectangles(4, [[-7.0, -3.0], [-3.0, -7.0]], [[2.0, 1.0], [2.0, 1.0]], -np.pi / 4)
        kw = dict(project_x=cs, project_u=Box(cfg["u_lo"], cfg["u_hi"]), max_line_search_iter=20, max_admm_iter=5,
                  rho_x=rho_x, rho_u=cfg["rho_u"], alpha=1.0, tol=0.0)
    s = _make(cfg, range(B))
    s.ilqr_admm(max_iter=1, **kw)                              # warm-up (allocations, first-call initialisation)
    s = _make(cfg, range(B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.ilqr_admm(max_iter=a.outer, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    done = len(s.cost_log) - 1 if hasattr(s, "cost_log") else a.outer
    done = max(1, min(done, a.outer))
    print(json.dumps({"config": a.config, "batch": B, "horizon": cfg["N"], "x_dim": cfg["n"], "u_dim": cfg["m"],
                      "admm_iters_J": kw["max_admm_iter"], "line_search_L": kw["max_line_search_iter"],
                      "outer_iterations": a.outer, "ms_per_outer_iteration": 1e3 * dt / a.outer,
                      "iterations_per_s": a.outer / dt, "cost_median": float(np.median(np.asarray(s.cost, dtype=np.float64)))}))


if __name__ == "__main__":
    main()
