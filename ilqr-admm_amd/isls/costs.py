"""Built-in non-quadratic cost models with a device implementation (line search + expansion).

The reference takes arbitrary `cost_function(x, u)` / `get_Cs(x, u)` callbacks (isls/isls.py:102,360); a HIP kernel
cannot call back into Python, so a cost that the line search evaluates for every candidate has to be one the kernels
know.  `PseudoHuber` is the car-parking cost of notebooks/Tutorial.ipynb cell 14 (Tassa et al.): assigning an instance to
`iSLS.cost_function` selects ISLS_COST_PHUBER on the device; the object itself is the numpy version of the same cost
(and of its derivatives, which the notebook gets from autograd) for use on the host and in tests.
"""
import numpy as np

from . import _capi as capi


class PseudoHuber:
    """sum_t [ sum_i cu_i u_ti^2 + sum_i cx_i ph(x_ti, px_i) ] + sum_i cf_i ph(x_{N-1,i}, pf_i),  ph(x,p) = sqrt(x^2+p^2) - p.
    cu [m]; cx, px, cf, pf [n] (entries of cx / cf may be zero; the matching px / pf must not be)."""
    cost_model = capi.COST_PHUBER

    def __init__(self, cu, cx, px, cf, pf):
        self.cu, self.cx, self.px, self.cf, self.pf = (np.asarray(v, dtype=np.float64).reshape(-1) for v in (cu, cx, px, cf, pf))

    def params(self):
        return np.concatenate([self.cu, self.cx, self.px, self.cf, self.pf])

    @staticmethod
    def _ph(x, p):
        return np.sqrt(x ** 2 + p ** 2) - p

    def __call__(self, x, u):
        """x [..., N, n], u [..., N, m] -> cost [...] (Tutorial.ipynb cell 14, `cost`; NaN -> 1e6 for stacked candidates)."""
        x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
        c = np.sum(self.cu * u ** 2, axis=(-1, -2)) + np.sum(self.cx * self._ph(x, self.px), axis=(-1, -2))
        c = c + np.sum(self.cf * self._ph(x[..., -1, :], self.pf), axis=-1)
        if x.ndim == 3:
            c = np.where(np.isnan(c), 1e6, c)
        return c

    def get_Cs(self, x, u):
        """(cs [N, n+m], Cs [N, n+m, n+m]): gradient and Hessian per time step (Tutorial.ipynb cell 16 without autograd)."""
        x, u = np.asarray(x, dtype=np.float64), np.asarray(u, dtype=np.float64)
        N, n = x.shape
        m = u.shape[1]
        s1 = np.sqrt(x ** 2 + self.px ** 2)
        g, h = self.cx * x / s1, self.cx * self.px ** 2 / s1 ** 3
        s2 = np.sqrt(x[-1] ** 2 + self.pf ** 2)
        g[-1] += self.cf * x[-1] / s2
        h[-1] += self.cf * self.pf ** 2 / s2 ** 3
        cs = np.concatenate([g, 2 * self.cu * u], axis=1)
        Cs = np.zeros((N, n + m, n + m))
        idx = np.arange(n + m)
        Cs[:, idx, idx] = np.concatenate([h, np.broadcast_to(2 * self.cu, (N, m))], axis=1)
        return cs, Cs
