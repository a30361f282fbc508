"""Drop-in for ``vittles/solver_lib.py`` with the solves on the MI355X.

Same four entry points, same argument meaning, same error / warning behaviour
(reference: /root/reference/vittles/solver_lib.py:7-98).  Every closure returns
the container type it was given: numpy in -> numpy out, CUDA tensor in -> CUDA
tensor out.  There is no CPU fallback.
"""
import warnings

import numpy as np
import scipy.sparse
import torch

from . import device as dev
from .linalg import DeviceCholesky


def get_dense_cholesky_solver(h, h_chol=None):
    """Return ``solve(v) = h^{-1} v`` via a dense Cholesky factorisation on the GPU
    (reference solver_lib.py:7-30).  ``h_chol`` may be a ``DeviceCholesky`` or the
    ``(c, lower)`` tuple of ``scipy.linalg.cho_factor``; if given, ``h`` is ignored."""
    dev.require_gpu()
    if h_chol is None:
        chol = DeviceCholesky.factor(dev.to_device(h))
    elif isinstance(h_chol, DeviceCholesky):
        chol = h_chol
    else:
        chol = DeviceCholesky.from_scipy(h_chol)

    def solve(v):
        x = chol.solve(dev.to_device(v, chol.L.device))
        return dev.like_input(x, v)
    solve.cholesky = chol
    return solve


def get_sparse_cholesky_solver(h):
    """Return ``solve(v) = h^{-1} v`` for a sparse ``h`` (reference solver_lib.py:33-48).
    Block-arrowhead matrices made by ``SparseBlockHessian`` keep their structure and
    are solved by the batched block kernels; any other sparse matrix is densified
    onto the device (the reference hands it to SuperLU)."""
    if not scipy.sparse.issparse(h):
        raise ValueError('`h` must be sparse.')
    solver = getattr(h, '_vt_solver', None)
    if solver is not None:
        return solver()
    return get_dense_cholesky_solver(np.asarray(h.todense()))


def get_cholesky_solver(h):
    """Dense or sparse dispatch (reference solver_lib.py:51-67)."""
    if scipy.sparse.issparse(h):
        return get_sparse_cholesky_solver(h)
    return get_dense_cholesky_solver(h)


def get_cg_solver(mat_times_vec, dim, cg_opts={}):
    """Conjugate-gradient ``solve(v)`` (reference solver_lib.py:70-98)."""
    from .cg import make_cg_solver
    return make_cg_solver(mat_times_vec, dim, cg_opts)
