#!/usr/bin/env python3
"""Per-kernel timing of the headline workload's launches (config 2, B=4096, N=100, fp64), each kernel on its own:
    python tools/kbench.py [--reps 40] [--batch 4096] [--libs ab/x/libisls_hip.so,ab/y/libisls_hip.so]
With --libs the script re-runs itself once per library (ISLS_HIP_LIB) as a child process and prints one line per library."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "ilqr-admm_amd"), ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def run(args):
    import numpy as np
    import torch
    import isls_problems as P
    from isls import models
    from isls.engine import Engine
    B, N, J, L = args.batch, 100, 5, 20
    dev = torch.device("cuda", 0)
    cfg = P.config2(batch=B, N=N, seed=0)
    n, m = cfg["n"], cfg["m"]
    eng = Engine(B, N, n, m, dtype=torch.float64, device=dev)
    mdl = models.LTI(cfg["A"], cfg["B"])
    eng.set_model(mdl.model_id, mdl.params())
    eng.set_quadratic_cost(cfg["zs"], cfg["Qs"], cfg["seq"], cfg["u_std"])
    eng.set_nominal(np.repeat(cfg["x0"][:, None, :], N, axis=1), cfg["u0"])
    eng.set_admm(rho_u=cfg["rho_u"], u_box=(cfg["u_lo"], cfg["u_hi"]), relax=cfg["relax"])
    eng.build_outer(L, J, tol_abs=0.0, tol_rel=0.0)
    for _ in range(2):                                         # a valid state for every kernel
        eng.linearize(); eng.expand(); eng.run_outer(); eng.accept_x_step()
    rec, seg = eng.ff_record(), eng.ff_seg()
    act = eng.admm_active

    def timed(fn, reps):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3               # us

    out = {}
    out["gain"] = timed(lambda: eng.gain(active=act, rec=rec, seg=seg), args.reps)
    out["ff"] = timed(lambda: eng.feedforward(active=act, seg=seg, rec=rec), args.reps)
    out["prep"] = timed(lambda: eng.feedforward_prepare(seg, active=act, rec=rec), args.reps) if seg is not None else 0.0
    out["rollout"] = timed(lambda: eng.rollout(L, active=act), args.reps)
    out["lin"] = timed(eng.linearize, args.reps)
    out["expand"] = timed(eng.expand, args.reps)
    out["run_outer"] = timed(eng.run_outer, max(5, args.reps // 4))
    import ctypes
    from isls.engine import library
    lib = library()
    lib.isls_timing_create.restype = ctypes.c_void_p
    lib.isls_timing_read_ms.restype = ctypes.c_double

    def families(fn, tag):
        """per-family kernel time inside run_outer (HIP events of the C driver) while `fn` loops"""
        tm = ctypes.c_void_p(lib.isls_timing_create())
        eng._outer_args.timing = tm
        lib.isls_timing_reset(tm)
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        for kind, name in ((0, "gain"), (1, "ff"), (2, "rollout")):
            cnt = ctypes.c_int(0)
            ms = lib.isls_timing_read_ms(tm, kind, ctypes.byref(cnt))
            out[f"{tag}.{name}"] = ms / max(1, cnt.value) * 1e3
        eng._outer_args.timing = None
        lib.isls_timing_destroy(tm)

    def outer():
        eng.linearize(); eng.expand(); eng.run_outer(); eng.accept_x_step(); eng.reduce()
    out["outer_it"] = timed(outer, max(5, args.reps // 4))
    families(outer, "A")
    if hasattr(eng, "advance"):                                # end of iteration + start of the next in one launch
        eng.build_outer(L, J, tol_abs=0.0, tol_rel=0.0, begin_done=True)
        eng.begin_outer()
        out["advance"] = timed(eng.advance, args.reps)
        def outer2():
            eng.run_outer(); eng.advance(); eng.reduce()
        out["outer_it2"] = timed(outer2, max(5, args.reps // 4))
        families(outer2, "B")
    print(json.dumps({k: round(v, 1) for k, v in out.items()}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--libs", default="")
    a = ap.parse_args()
    if a.libs:
        for lib in a.libs.split(","):
            env = dict(os.environ)
            if lib != "default":
                env["ISLS_HIP_LIB"] = os.path.join(ROOT, lib)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--reps", str(a.reps), "--batch", str(a.batch)],
                               env=env, capture_output=True, text=True)
            print(lib, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "FAILED " + r.stderr[-400:])
    else:
        run(a)
