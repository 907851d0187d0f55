"""Helpers of the reference's `isls.utils` that the hot path and the notebooks use:
`get_double_integrator_AB`, `find_mus`, `find_precs`.  Plotting / spline / null-space helpers are out of scope."""
from math import factorial

import numpy as np


def get_double_integrator_AB(nb_dim, nb_deriv=2, dt=0.01):
    """A = kron(A1d, I), B = kron(B1d, I) with A1d[i,i+j] = dt^j/j!, B1d[nb_deriv-j] = dt^j/j!  (isls/utils.py:266-276)."""
    A1 = sum(np.diag(np.full(nb_deriv - j, dt ** j / factorial(j)), j) for j in range(nb_deriv))
    B1 = np.array([[dt ** (nb_deriv - r) / factorial(nb_deriv - r)] for r in range(nb_deriv)])
    eye = np.eye(nb_dim)
    return np.kron(A1, eye), np.kron(B1, eye)


def find_mus(zs, seq):
    """Stacked via-point targets xd = [zs[seq[0]], zs[seq[1]], ...]  (isls/utils.py:95-99)."""
    return np.concatenate([np.asarray(zs)[s] for s in seq])


def find_precs(Qs, seq):
    """Per-timestep precision blocks Q_t = Qs[seq[t]] as an [N,n,n] array (the reference builds the
    block-diagonal sparse matrix of them, isls/utils.py:101-115; only the blocks are ever used)."""
    return np.stack([np.asarray(Qs)[s] for s in seq])
