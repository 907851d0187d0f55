"""DualKernels: run every kernel call on the CPU oracle (numpy) AND on the HIP library (torch, cuda:0)
with identical inputs and compare every array afterwards.  Test infrastructure only."""
import numpy as np
import torch

from isls import _capi as capi


class DualKernels:
    def __init__(self, oracle, hip, tol=1e-10, int_exact=True, verbose=False, ff_nseg=1, ff_record=False, ti_weights=False,
                 ff_lin=False):
        self.oracle, self.hip, self.tol, self.int_exact, self.verbose = oracle, hip, tol, int_exact, verbose
        # True: ADMM weights that are the same at every step ([N,d,d] / [N,d] tiles of one block, what compute_Rr_Qr builds from
        # a scalar rho) reach the HIP side as ONE block ([1,d,d] / [1,d]: time stride 0), the way isls.Engine hands them over --
        # the record feed-forward pass then takes its one-hand-off form (riccati_ffrec2_kernel) and the rollout keeps the AL
        # weights in its record instead of loading them per step; the oracle still gets the tiled arrays
        self.ti_weights = ti_weights
        # > 1: the HIP side runs the feed-forward pass in its time-parallel form (isls_ffseg: prepare + segmented
        # recursion + stitch) while the oracle keeps the reference's sequential recursion
        self.ff_nseg = ff_nseg
        # True: the HIP gain pass also writes the packed step records (isls_gain_args.rec, NaN-filled before) and the HIP
        # feed-forward passes read those instead of A, B, K, Quu, fac, Qux; the oracle keeps the reference's recursion
        self.ff_record, self._rec = ff_record, None
        # True: the HIP record passes get the hint isls.Engine gives them (isls_ff_args.lin_on) when A, B came from `linearize`
        # of a double integrator or the 3R arm: [K | fac] of the records + the model's structure instead of the dense
        # [Phi | B] blocks; `lin_calls` counts the passes that ran with it
        self.ff_lin, self._lin, self.lin_calls = ff_lin, None, 0
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
        reexpressed = set()
        if self.ti_weights:
            for key in ("Qr", "Rr", "wq", "wr"):
                w = dkw.get(key)
                if w is not None and w.ndim == (3 if key in ("Qr", "Rr") else 2) and w.shape[0] > 1 and bool((w == w[:1]).all()):
                    dkw[key] = w[:1].contiguous()
                elif (key == "Qr" and name == "riccati_ff" and self.ff_record and self.ff_nseg <= 1 and w is not None and w.ndim == 3
                      and w.shape[0] > 2 and bool((w[:-1] == w[:1]).all())):
                    # the same block at every step but the terminal one (a terminal state constraint): isls.Engine hands that
                    # over as one block + isls_ff_args.Qr_term, which keeps the one-hand-off record kernel
                    dkw["Qr"], dkw["Qr_term"] = w[:1].contiguous(), w[-1].contiguous()
                    reexpressed.add("Qr")                      # an input in another form: nothing to compare afterwards
        if name == "riccati_ff" and self.ff_nseg > 1:
            dkw = dict(dkw, seg=self._prepare_segments(dargs, dkw))
        if self.ff_record and name == "riccati_gain":
            B, N, m, n = dargs[4].shape
            self._rec = torch.full((capi.ff_record_elems(B, N, n, m),), float("nan"), dtype=dargs[4].dtype, device="cuda")
            self._rec_dims = (B, N)
            dkw = dict(dkw, rec=self._rec)
        lean_gain = None
        if self.ff_record and name == "riccati_gain" and self.ff_lin and self._lin is not None:
            # the structured passes read the LEAN records a gain pass with the same hint writes (record form without the Quu /
            # fac / Qux arrays): a second gain launch on a copy of K, behind the one that is compared with the oracle
            self._rec_lean = torch.full_like(self._rec, float("nan"))
            lean_gain = (list(dargs[:4]) + [dargs[4].clone(), None, None, None],
                         dict({k: v for k, v in dkw.items() if k != "rec"}, rec=self._rec_lean, lin=self._lin,
                              status=torch.zeros_like(dkw["status"]) if dkw.get("status") is not None else None))
        if self.ff_record and name == "riccati_ff" and self._rec is not None and self._rec_dims == tuple(dargs[4].shape[:2]):
            dkw = dict(dkw, rec=self._rec)
            if self.ff_lin and self._lin is not None and getattr(self, "_rec_lean", None) is not None:
                dkw = dict(dkw, rec=self._rec_lean, lin=self._lin)
                self.lin_calls += 1
        if name == "linearize":                                # what the engine knows about its A, B (Engine.ff_lin)
            self._lin = self._lin_hint(args[0], args[1], dargs[1])
        for blk in ("x", "u"):                                 # set descriptors hold pointers: rebuild them on the device
            if dkw.get(blk + "_sets") is not None:
                dkw[blk + "_sets"] = self._rebuild_sets(dkw[blk + "_sets"]._spec, dkw[blk + "_work"])
        getattr(self.hip, name)(*dargs, **dkw)
        if lean_gain is not None:
            self.hip.riccati_gain(*lean_gain[0], **lean_gain[1])
            torch.cuda.synchronize()
            k_dense, k_lean = dargs[4], lean_gain[0][4]
            rel = float((k_dense - k_lean).abs().max() / max(1.0, float(k_dense.abs().max())))
            if rel > self.tol:                                  # (another instance of the kernel than the array-writing one above)
                raise AssertionError(f"riccati_gain with the hint: K differs from the dense pass by {rel:.3e}")
        torch.cuda.synchronize()
        self.calls += 1
        for i, (h, d) in enumerate(zip(args, dargs)):
            if isinstance(h, np.ndarray):
                self._compare(name, f"arg{i}", h, d)
        for k in kw:
            if isinstance(kw[k], np.ndarray) and k not in reexpressed:
                self._compare(name, k, kw[k], dkw[k])

    @staticmethod
    def _lin_hint(model, par, par_dev):
        from isls import models
        if model in (capi.MODEL_ARM3R, capi.MODEL_CAR):
            return (model, par_dev)
        if model == capi.MODEL_DI:
            return (capi.MODEL_DI, par_dev)
        if model == capi.MODEL_LTI:                            # isls.models.LTI recognises a double integrator in a dense pair
            p = np.asarray(par, dtype=np.float64)
            for n in range(2, 17, 2):                          # par = [A (n x n), B (n x n/2)]
                if p.size == n * n + n * (n // 2):
                    mdl = models.LTI(p[:n * n].reshape(n, n), p[n * n:].reshape(n, n // 2))
                    if mdl.model_id == capi.MODEL_DI:
                        return (capi.MODEL_DI, torch.as_tensor(np.asarray(mdl.params(), dtype=par.dtype)).cuda())
        return None

    def _rebuild_sets(self, spec, work):
        """The same descriptor (and its further stages) over device copies of the operands."""
        nxt = self._rebuild_sets(spec["next_stage"]._spec, work) if spec.get("next_stage") is not None else None
        dsets = [{k: self._to_dev(v) for k, v in st.items()} for st in spec["sets"]]
        return capi.Kernels.project_args(work, work, dsets, rho=spec["rho"], max_iter=spec["max_iter"], threshold=spec["threshold"],
                                         cols=spec["cols"], algorithm=spec.get("algorithm", 0), row_mask=self._to_dev(spec.get("row_mask")),
                                         next_stage=nxt)

    def _prepare_segments(self, dargs, dkw):
        """Fresh NaN-filled operator buffers + isls_riccati_ff_prepare on the device operands of an ff call."""
        A, Bm, _, _, K, Quu, fac, Qux = dargs[:8]
        B, N, m, n = K.shape
        nseg, seg_len = self.hip.ff_segments(N, self.ff_nseg)
        if nseg < 2:
            return None
        nan = lambda *shape: torch.full(shape, float("nan"), dtype=K.dtype, device=K.device)   # noqa: E731
        self._seg_bufs = (nan(B, N, m, n), nan(B, nseg, n, n), nan(B, nseg, n))
        seg = capi.Kernels.ff_seg(*self._seg_bufs, seg_len)
        rec = self._rec if (self.ff_record and self._rec is not None and self._rec_dims == (B, N)) else None
        self.hip.riccati_ff_prepare(A, Bm, K, Quu, fac, Qux, seg, solve_mode=dkw.get("solve_mode", capi.SOLVE_CHOL),
                                    active=dkw.get("active"), rec=rec)
        return seg

    def __getattr__(self, name):
        if name in ("riccati_gain", "riccati_ff", "rollout_ls", "admm_update", "expand_quadratic", "linearize", "accept_step"):
            return lambda *a, **kw: self._dual(name, a, kw)
        raise AttributeError(name)


def hip_kernels():
    return capi.Kernels(capi.load_hip_library(), prefix="isls_", with_stream=True)
