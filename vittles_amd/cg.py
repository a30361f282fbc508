"""Conjugate-gradient solver on the device (K6): the drop-in behind
``solver_lib.get_cg_solver`` (reference vittles/solver_lib.py:70-98).

Semantics follow what the reference obtains from ``scipy.sparse.linalg.cg(linop, v,
**cg_opts, atol='legacy')`` (SciPy <= 1.11): x0 = 0 unless given, stop when
``||r|| < tol * ||b||`` (tol = cg_opts['tol'] or ['rtol'], default 1e-5), at most
``maxiter`` (default 10 * dim) iterations, no preconditioner; on exhaustion a
``UserWarning('CG exited with error code {maxiter}')`` is raised and x is returned anyway.
"""
import warnings

import numpy as np
import torch

from . import _lib, device as dev, rows


class GLMHessianVectorProduct:
    """Registered ``mat_times_vec``: v -> X^T (d .* (X v)) + prior .* v on the device (fused
    HVP kernel, X read once per product; partial products all-reduced when row-sharded).
    ``apply_multi`` takes K vectors as the rows of a (K, D) tensor: one pair of MFMA passes over X
    for all of them (``rows.glm_hvp_multi``) from ``MULTI_MIN`` vectors on, a loop of single products below."""

    MULTI_MIN = 4      # two tile passes over X cost about as much as 3-4 single-vector products

    def __init__(self, objective, opt_par, hyper_par):
        self.obj = objective
        _, _, d = rows.glm_link(objective, opt_par, hyper_par)
        self.d = d
        self.dim = objective.D
        self.device = objective.device

    def apply(self, v, out=None):
        obj = self.obj
        prior = obj.prior if obj._rank() == 0 else None
        out = rows.glm_hvp(obj.X, self.d, v, prior=prior, out=out)
        obj._all_reduce(out)
        return out

    def apply_multi(self, V, out=None):
        obj = self.obj
        K = V.shape[0]
        if out is None:
            out = torch.empty((K, self.dim), dtype=torch.float64, device=V.device)
        prior = obj.prior if obj._rank() == 0 else None
        if K >= self.MULTI_MIN and K <= 128 and obj.X.dtype == torch.float64:
            rows.glm_hvp_multi(obj.X, self.d, V, prior=prior, out=out)
        else:
            for k in range(K):
                rows.glm_hvp(obj.X, self.d, V[k], prior=prior, out=out[k])
        obj._all_reduce(out)          # ONE message for the K partial products
        return out

    def __call__(self, v):
        vd = dev.to_device(v, self.device).contiguous()
        return dev.like_input(self.apply(vd), v)


def cg_solve_multi(matvec_multi_dev, B, X0=None, tol=1e-5, maxiter=None, minv=None, callback=None):
    """K right-hand sides in lock-step: the rows of the (K, D) CUDA tensor ``B``.  Every column runs the recurrence
    of ``scipy.sparse.linalg.cg`` (legacy stopping rule: ``||r_k|| < tol * ||b_k||``) on its own and is frozen once
    it has converged, so each solution is what a single-vector solve returns.  ``matvec_multi_dev(P, Q)`` fills
    ``Q[k] = A P[k]``; ``minv``: the diagonal of a preconditioner M ~ A^{-1} as a (D,) CUDA vector; ``callback(k,
    x_k)`` after every iteration of a still-active column (x_k: a CUDA view).  Returns (X, info (K,), iterations
    (K,)); info is 0, ``maxiter`` (exhausted, as SciPy reports it) or -1 (breakdown: ``p.Ap <= 0`` -- the operator is
    not positive definite -- or a non-finite residual; the column is frozen at the iterate before it)."""
    lib = dev.require_gpu()
    K, D = B.shape
    if maxiter is None:
        maxiter = D * 10
    st = dev.stream_ptr()
    B = B.contiguous()
    X = torch.zeros((K, D), dtype=torch.float64, device=B.device) if X0 is None else X0.clone().contiguous()
    R, P, Q = torch.empty_like(X), torch.empty_like(X), torch.empty_like(X)
    scal = torch.zeros((K, 8), dtype=torch.float64, device=B.device)
    Q0 = None
    if X0 is not None and bool(X0.any()):
        Q0 = torch.empty_like(X)
        matvec_multi_dev(X, Q0)
    _lib.check(lib.vt_cg_init_multi(B.data_ptr(), dev.ptr(Q0), R.data_ptr(), P.data_ptr(), D, D, K, dev.ptr(minv),
                                    scal.data_ptr(), st), "vt_cg_init_multi")
    h = scal.cpu().numpy()
    rnorm, bnorm = h[:, 1].copy(), h[:, 4]
    # SciPy's legacy tolerance (_get_atol): tol * ||b||, or tol itself for a zero right-hand side; a column whose
    # first residual is already <= tol returns its x0 untouched
    atol = np.where(bnorm == 0, tol, tol * bnorm)
    atol = np.where(rnorm <= tol, np.inf, atol)
    scal[:, 6] = torch.as_tensor(atol, device=B.device)
    info = np.zeros(K, dtype=np.int64)
    broken = np.zeros(K, dtype=bool)         # breakdown: p.Ap <= 0 (A is not positive definite) or a non-finite residual
    active = ~(rnorm < atol)
    for _ in range(maxiter):
        active = ~(rnorm < atol) & ~broken
        if not active.any():
            break
        matvec_multi_dev(P, Q)
        _lib.check(lib.vt_cg_update_multi(P.data_ptr(), Q.data_ptr(), X.data_ptr(), R.data_ptr(), D, D, K,
                                          dev.ptr(minv), scal.data_ptr(), st), "vt_cg_update_multi")
        h = scal.cpu().numpy()               # the iteration's only host sync (64 K bytes)
        rnorm = h[:, 1].copy()
        bad = active & (~np.isfinite(rnorm) | ~(h[:, 3] > 0.0))
        if bad.any():
            broken |= bad
            scal[:, 6] = torch.as_tensor(np.where(broken, np.inf, atol), device=B.device)   # freeze those columns
        if callback is not None:
            for k in np.nonzero(active)[0]:
                callback(int(k), X[k])
    else:
        # SciPy's for-else: a column that was still iterating in the last pass reports maxiter, converged or not
        info[active] = maxiter
    info[broken] = -1
    iters = scal[:, 5].cpu().numpy().astype(np.int64)
    return X, info, iters


def cg_solve(matvec_dev, b, x0=None, tol=1e-5, maxiter=None, minv=None, callback=None):
    """Run CG with device vectors.  ``matvec_dev(p, out)`` fills ``out`` = A p (CUDA tensors).
    Returns (x, info, iterations)."""
    D = b.shape[0]
    X, info, iters = cg_solve_multi(lambda P, Q: matvec_dev(P[0], Q[0]), b.reshape(1, D),
                                    X0=None if x0 is None else x0.reshape(1, D), tol=tol, maxiter=maxiter, minv=minv,
                                    callback=None if callback is None else (lambda k, xk: callback(xk)))
    return X[0], int(info[0]), int(iters[0])


def _diagonal_preconditioner(M, dim, device):
    """SciPy's ``M`` (an approximation of A^{-1} applied as ``M @ r``) when it is DIAGONAL: a (dim,) vector, or a
    (dim, dim) array / scipy.sparse matrix with nothing off the diagonal.  Anything else would have to run on the
    host between two kernels of every iteration."""
    if M is None:
        return None
    if isinstance(M, torch.Tensor) or isinstance(M, np.ndarray):
        if M.ndim == 1:
            if M.shape[0] != dim:
                raise ValueError('M must have %d entries' % dim)
            return dev.to_device(M, device).contiguous()
        Mh = M.cpu().numpy() if isinstance(M, torch.Tensor) else M
        if Mh.shape == (dim, dim) and not (Mh - np.diag(np.diag(Mh))).any():
            return dev.to_device(np.ascontiguousarray(np.diag(Mh)), device)
    elif hasattr(M, 'diagonal') and hasattr(M, 'nnz') and M.shape == (dim, dim):
        diag = np.asarray(M.diagonal(), dtype=np.float64)
        if M.nnz == np.count_nonzero(diag) or abs(M - type(M)(np.diag(diag))).nnz == 0:
            return dev.to_device(diag, device)
    raise NotImplementedError(
        'cg_opts["M"]: only a DIAGONAL preconditioner is supported on the device -- pass its diagonal as a '
        '(%d,) vector, or a diagonal (%d, %d) array / scipy.sparse matrix' % (dim, dim, dim))


def make_cg_solver(mat_times_vec, dim, cg_opts={}):
    """``cg_opts`` as the reference forwards them to SciPy (vittles/solver_lib.py:93): ``tol`` / ``rtol``,
    ``maxiter``, ``x0``, ``M`` (diagonal, see above), ``callback`` (called with the current iterate after every
    iteration, in the container type of the right-hand side); ``atol`` is ignored (the reference always passes
    'legacy').  The returned solver takes a (dim,) vector or -- build-side extension -- a (dim, K) matrix of K
    right-hand sides, solved in lock-step with one multi-vector product per iteration."""
    dev.require_gpu()
    opts = dict(cg_opts)
    tol = opts.pop('tol', opts.pop('rtol', 1e-5))
    maxiter = opts.pop('maxiter', None)
    x0 = opts.pop('x0', None)
    M = opts.pop('M', None)
    callback = opts.pop('callback', None)
    opts.pop('atol', None)
    if opts:
        raise TypeError('unsupported cg_opts: %s' % sorted(opts))
    registered = hasattr(mat_times_vec, 'apply')

    def matvec_dev(p, out):
        if registered:
            mat_times_vec.apply(p, out)
        else:
            # user callable (numpy in, numpy out) as in the reference's LinearOperator
            res = mat_times_vec(p.cpu().numpy())
            out.copy_(torch.as_tensor(np.asarray(res, dtype=np.float64).reshape(-1)))

    def matvec_multi_dev(P, Q):
        if registered and hasattr(mat_times_vec, 'apply_multi'):
            mat_times_vec.apply_multi(P, Q)
        else:
            for k in range(P.shape[0]):
                matvec_dev(P[k], Q[k])

    def solve(v):
        b = dev.to_device(v).contiguous()
        if b.ndim not in (1, 2) or b.shape[0] != dim:
            raise ValueError('shapes of A ({0}, {0}) and b {1} are incompatible'.format(dim, tuple(b.shape)))
        minv = _diagonal_preconditioner(M, dim, b.device)
        x0d = None if x0 is None else dev.to_device(x0, b.device).contiguous()
        if b.ndim == 1:
            cb = None if callback is None else (lambda xk: callback(dev.like_input(xk.clone(), v)))
            x, info, _ = cg_solve(matvec_dev, b, x0=x0d, tol=tol, maxiter=maxiter, minv=minv, callback=cb)
            if info != 0:
                warnings.warn('CG exited with error code {}'.format(info))
            return dev.like_input(x, v)
        # (dim, K): K recurrences in lock-step; vectors travel as the rows of (K, dim) arrays
        Bk = b.t().contiguous()
        X0k = None
        if x0d is not None:
            X0k = (x0d.t() if x0d.ndim == 2 else x0d.reshape(1, dim).expand(Bk.shape[0], dim)).contiguous()
        cb = None if callback is None else (lambda k, xk: callback(dev.like_input(xk.clone(), v)))
        Xk, infos, _ = cg_solve_multi(matvec_multi_dev, Bk, X0=X0k, tol=tol, maxiter=maxiter, minv=minv, callback=cb)
        for code in infos[infos != 0][:1]:
            warnings.warn('CG exited with error code {}'.format(int(code)))
        return dev.like_input(Xk.t(), v)
    return solve
