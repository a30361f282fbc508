"""ORACLE (test infrastructure, never a product path): restatement of
``/root/reference/vittles/solver_lib.py`` on the CPU.

The dense and sparse closures call the same SciPy entry points as the reference
(LAPACK potrf/potrs via cho_factor/cho_solve, SuperLU via factorized).  The CG
closure cannot call ``scipy.sparse.linalg.cg(..., atol='legacy')`` as the
reference does (solver_lib.py:93) because SciPy >= 1.14 rejects that value, so
the iteration is restated from the published SciPy 1.11 algorithm
(``scipy/sparse/linalg/_isolve/iterative.py``: ``cg`` + ``_get_atol``):
"parity unpinned" for CG beyond the reference's own test (agreement with
``np.linalg.solve`` to 6 decimals, tests/test_solver_lib.py:33-43).
"""
import warnings

import numpy as np
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg


def get_dense_cholesky_solver(h, h_chol=None):
    # solver_lib.py:7-30
    if h_chol is None:
        h_chol = scipy.linalg.cho_factor(h)

    def solve(v):
        return scipy.linalg.cho_solve(h_chol, v)
    return solve


def get_sparse_cholesky_solver(h):
    # solver_lib.py:33-48
    if not scipy.sparse.issparse(h):
        raise ValueError('`h` must be sparse.')
    return scipy.sparse.linalg.factorized(h)


def get_cholesky_solver(h):
    # solver_lib.py:51-67
    if scipy.sparse.issparse(h):
        return get_sparse_cholesky_solver(h)
    return get_dense_cholesky_solver(h)


def cg_legacy(matvec, b, x0=None, tol=1e-5, maxiter=None, return_iters=False):
    """SciPy <= 1.11 ``cg(A, b, tol=tol, atol='legacy')`` without preconditioner.

    _get_atol('legacy'): if ||b - A x0|| <= tol -> exit at once; atol = tol*||b||
    (tol if ||b|| == 0).  Loop (maxiter = 10 n): if ||r|| < atol -> (x, 0);
    rho = r.r; p = r + (rho/rho_prev) p; q = A p; alpha = rho / p.q;
    x += alpha p; r -= alpha q.  Exhausted -> (x, maxiter).
    """
    b = np.asarray(b, dtype=np.float64)
    n = b.shape[0]
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64)
    if maxiter is None:
        maxiter = n * 10
    bnrm2 = np.linalg.norm(b)
    r = b - matvec(x) if x.any() else b.copy()
    if np.linalg.norm(r) <= tol:
        return (x, 0, 0) if return_iters else (x, 0)
    atol = tol if bnrm2 == 0 else tol * float(bnrm2)
    rho_prev, p = None, None
    for iteration in range(maxiter):
        if np.linalg.norm(r) < atol:
            return (x, 0, iteration) if return_iters else (x, 0)
        rho_cur = np.dot(r, r)
        if iteration > 0:
            beta = rho_cur / rho_prev
            p *= beta
            p += r
        else:
            p = r.copy()
        q = matvec(p)
        alpha = rho_cur / np.dot(p, q)
        x += alpha * p
        r -= alpha * q
        rho_prev = rho_cur
    return (x, maxiter, maxiter) if return_iters else (x, maxiter)


def get_cg_solver(mat_times_vec, dim, cg_opts={}):
    # solver_lib.py:70-98 (warning text and "return x regardless" kept)
    opts = dict(cg_opts)
    tol = opts.pop('tol', opts.pop('rtol', 1e-5))
    maxiter = opts.pop('maxiter', None)
    x0 = opts.pop('x0', None)

    def solve(v):
        x, info = cg_legacy(mat_times_vec, v, x0=x0, tol=tol, maxiter=maxiter)
        if info != 0:
            warnings.warn('CG exited with error code {}'.format(info))
        return x
    return solve
