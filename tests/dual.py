"""DualKernels: run every kernel call on the CPU oracle (numpy) AND on the HIP library (torch, cuda:0)
with identical inputs and compare every array afterwards.  Test infrastructure only."""
import numpy as np
import torch

from isls import _capi as capi


class DualKernels:
    def __init__(self, oracle, hip, tol=1e-10, int_exact=True, verbose=False):
        self.oracle, self.hip, self.tol, self.int_exact, self.verbose = oracle, hip, tol, int_exact, verbose
        self.max_err = {}
        self.calls = 0

    def _to_dev(self, x):
        if isinstance(x, np.ndarray):
            # keep broadcast (stride-0) structure out of the picture: dense device copy of the same shape
            return torch.from_numpy(np.ascontiguousarray(x)).cuda()
        return x

    def _compare(self, name, key, h, d):
        d = d.cpu().numpy()
        if h.dtype.kind in "iu":
            if self.int_exact and not np.array_equal(h, d):
                raise AssertionError(f"{name}:{key}: integer mismatch oracle={h[:8]} hip={d[:8]}")
            return
        fin = np.isfinite(h)
        scale = max(1.0, float(np.max(np.abs(h[fin]))) if fin.any() else 1.0)
        same = (np.isnan(h) & np.isnan(d)) | (h == d)               # identical infinities / NaNs agree
        with np.errstate(invalid="ignore"):
            diff = np.where(same, 0.0, np.abs(h - d))
        err = float(np.max(diff)) / scale if h.size else 0.0
        if not np.isfinite(err):
            err = float("inf")
        k = f"{name}:{key}"
        self.max_err[k] = max(self.max_err.get(k, 0.0), err)
        if err > self.tol:
            idx = np.unravel_index(np.argmax(diff), h.shape)
            raise AssertionError(f"{k}: rel err {err:.3e} > {self.tol:.1e} at {idx}: oracle={h[idx]!r} hip={d[idx]!r}")

    def _dual(self, name, args, kw):
        dargs = [self._to_dev(a) for a in args]
        dkw = {k: self._to_dev(v) for k, v in kw.items()}
        getattr(self.oracle, name)(*args, **kw)
        getattr(self.hip, name)(*dargs, **dkw)
        torch.cuda.synchronize()
        self.calls += 1
        for i, (h, d) in enumerate(zip(args, dargs)):
            if isinstance(h, np.ndarray):
                self._compare(name, f"arg{i}", h, d)
        for k in kw:
            if isinstance(kw[k], np.ndarray):
                self._compare(name, k, kw[k], dkw[k])

    def __getattr__(self, name):
        if name in ("riccati_gain", "riccati_ff", "rollout_ls", "admm_update", "expand_quadratic", "linearize", "accept_step"):
            return lambda *a, **kw: self._dual(name, a, kw)
        raise AttributeError(name)


def hip_kernels():
    return capi.Kernels(capi.load_hip_library(), prefix="isls_", with_stream=True)
