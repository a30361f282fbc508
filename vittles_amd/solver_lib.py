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
      * an unstructured matrix above that is solved ITERATIVELY on the device: block CG with a Jacobi preconditioner
        over a CSR product (``vt_spmv_csr``), relative residual ``SPARSE_CG_TOL`` -- not a factorisation, so a solve
        costs iterations, and a matrix CG cannot bring to that residual within 10 d iterations (not positive definite,
        or too ill-conditioned) raises ``LinAlgError`` instead of returning a poor answer.
    Every route needs a symmetric matrix with both triangles stored (checked, ``ValueError``).
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
        return _sparse_cg_solver(h)
    hd = np.asarray(h.todense())
    shl._require_symmetric(hd, hd.T, 'matrix')      # the dense Cholesky reads one triangle only
    return get_dense_cholesky_solver(hd)


SPARSE_CG_TOL = 1e-12       # relative residual of the iterative route for unstructured sparse matrices


class _CsrOperator:
    """Device CSR image of a symmetric sparse matrix: ``apply`` / ``apply_multi`` for the CG driver."""

    def __init__(self, csr, device):
        self.n = csr.shape[0]
        self.indptr = torch.as_tensor(csr.indptr.astype(np.int32), device=device)
        self.indices = torch.as_tensor(csr.indices.astype(np.int32), device=device)
        self.data = torch.as_tensor(np.asarray(csr.data, dtype=np.float64), device=device)

    def apply_multi(self, V, out=None):
        from . import _lib
        lib = dev.require_gpu()
        V = V if V.stride(-1) == 1 else V.contiguous()
        K = V.shape[0]
        if out is None:
            out = torch.empty((K, self.n), dtype=torch.float64, device=V.device)
        _lib.check(lib.vt_spmv_csr(self.indptr.data_ptr(), self.indices.data_ptr(), self.data.data_ptr(), self.n,
                                   V.data_ptr(), dev.ld(V), K, out.data_ptr(), dev.ld(out), dev.stream_ptr()),
                   "vt_spmv_csr")
        return out

    def apply(self, v, out=None):
        res = self.apply_multi(v.reshape(1, self.n), None if out is None else out.reshape(1, self.n))
        return res.reshape(self.n)


def _sparse_cg_solver(h):
    from . import sparse_hessian_lib as shl
    from .cg import cg_solve_multi
    csr = scipy.sparse.csr_matrix(h)
    csr.sum_duplicates()
    d = csr.shape[0]
    if csr.nnz >= 2 ** 31:
        raise ValueError('get_sparse_cholesky_solver: more than 2^31 stored entries')
    asym = abs(csr - csr.T)
    scale = abs(csr).max() if csr.nnz else 0.0
    if asym.nnz and asym.max() > shl.SYM_RTOL * scale:
        raise ValueError('get_sparse_cholesky_solver needs a SYMMETRIC (positive definite) matrix with both triangles '
                         'stored; max |h - h^T| = %.3g' % asym.max())
    diag = np.asarray(csr.diagonal(), dtype=np.float64)
    if not np.all(diag > 0):
        raise np.linalg.LinAlgError('matrix is not positive definite (non-positive diagonal entry)')
    device = torch.device('cuda', torch.cuda.current_device())
    op = _CsrOperator(csr, device)
    minv = torch.as_tensor(1.0 / diag, device=device)

    def solve(v):
        b = dev.to_device(v, device)
        if b.ndim not in (1, 2) or b.shape[0] != d:
            raise ValueError("right-hand side must have %d rows" % d)
        Bk = (b.reshape(d, 1) if b.ndim == 1 else b).t().contiguous()
        Xk, infos, _ = cg_solve_multi(op.apply_multi, Bk, tol=SPARSE_CG_TOL, maxiter=10 * d, minv=minv)
        if np.any(infos != 0):
            raise np.linalg.LinAlgError(
                'get_sparse_cholesky_solver: the iterative solve of the %d x %d unstructured sparse matrix did not reach '
                'a relative residual of %g (not positive definite, or too ill-conditioned for CG)' % (d, d, SPARSE_CG_TOL))
        out = Xk[0] if b.ndim == 1 else Xk.t()
        return dev.like_input(out, v)
    return solve


def get_cholesky_solver(h):
    """Dense or sparse dispatch (reference solver_lib.py:51-67)."""
    if scipy.sparse.issparse(h):
        return get_sparse_cholesky_solver(h)
    return get_dense_cholesky_solver(h)


def get_cg_solver(mat_times_vec, dim, cg_opts={}):
    """Conjugate-gradient ``solve(v)`` (reference solver_lib.py:70-98)."""
    from .cg import make_cg_solver
    return make_cg_solver(mat_times_vec, dim, cg_opts)
