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


DENSE_D_MAX = 8192          # largest D the dense device Cholesky covers (VT_DENSE_D_MAX)


def get_sparse_cholesky_solver(h, sparsity_array=None, global_inds=None):
    """Return ``solve(v) = h^{-1} v`` for a sparse ``h`` (reference solver_lib.py:33-48, which hands any
    sparse matrix to SuperLU).  Here the matrix must be symmetric positive definite (Cholesky, like the
    function's name says) and is solved on the device:

      * block-arrowhead matrices -- ``SparseBlockHessian`` results, and ANY scipy sparse matrix whose pattern is
        block-diagonal plus a dense border (``.tocsc()`` / ``.copy()`` / sums of such matrices, a COO a user
        assembled) -- by the batched block kernels; the structure is recovered from the sparsity pattern
        (``sparse_hessian_lib.detect_arrowhead``), or given by the build-side keywords ``sparsity_array`` (G, M)
        / ``global_inds``;
      * anything else up to ``DENSE_D_MAX`` rows is densified onto the device;
      * an unstructured matrix above that raises ``VittlesHipError`` (never a silent giant ``todense()``).
    """
    from . import sparse_hessian_lib as shl
    from ._lib import VittlesHipError
    if not scipy.sparse.issparse(h):
        raise ValueError('`h` must be sparse.')
    dev.require_gpu()
    d = h.shape[0]
    explicit = sparsity_array is not None or global_inds is not None
    struct = None if explicit else getattr(h, '_vt_structure', None)
    if struct is not None:
        # the attribute is a snapshot of device images: trust it only while the matrix still holds the values
        # it was taken from (an in-place edit of h.data keeps the attribute alive)
        if getattr(h, '_vt_fingerprint', None) != shl.matrix_fingerprint(h) or not struct.covers_everything() \
                or struct.M > shl.MAX_BLOCK or struct.Kg > shl.MAX_BLOCK:
            struct = None
    if struct is None and (explicit or d > 256):
        struct = shl.arrowhead_from_sparse(h, sparsity_array=sparsity_array, global_inds=global_inds)
    if struct is not None:
        return struct.make_solver()
    if d > DENSE_D_MAX:
        raise VittlesHipError(
            'get_sparse_cholesky_solver: the %d x %d matrix is not block-arrowhead (blocks and border of at most '
            '%d) and is too large for the dense device Cholesky (%d); pass sparsity_array= / global_inds= if the '
            'pattern alone does not show the structure.' % (d, d, shl.MAX_BLOCK, DENSE_D_MAX))
    hd = np.asarray(h.todense())
    shl._require_symmetric(hd, hd.T, 'matrix')      # the dense Cholesky reads one triangle only
    return get_dense_cholesky_solver(hd)


def get_cholesky_solver(h):
    """Dense or sparse dispatch (reference solver_lib.py:51-67)."""
    if scipy.sparse.issparse(h):
        return get_sparse_cholesky_solver(h)
    return get_dense_cholesky_solver(h)


def get_cg_solver(mat_times_vec, dim, cg_opts={}):
    """Conjugate-gradient ``solve(v)`` (reference solver_lib.py:70-98)."""
    from .cg import make_cg_solver
    return make_cg_solver(mat_times_vec, dim, cg_opts)
