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
    HVP kernel, X read once per product; partial products all-reduced when row-sharded)."""

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

    def __call__(self, v):
        vd = dev.to_device(v, self.device).contiguous()
        return dev.like_input(self.apply(vd), v)


def cg_solve(matvec_dev, b, x0=None, tol=1e-5, maxiter=None):
    """Run CG with device vectors.  ``matvec_dev(p, out)`` fills ``out`` = A p (CUDA tensors).
    Returns (x, info, iterations)."""
    lib = dev.require_gpu()
    D = b.shape[0]
    if maxiter is None:
        maxiter = D * 10
    st = dev.stream_ptr()
    x = torch.zeros(D, dtype=torch.float64, device=b.device) if x0 is None else x0.clone()
    r = torch.empty_like(x)
    p = torch.empty_like(x)
    q = torch.empty_like(x)
    scal = torch.zeros(8, dtype=torch.float64, device=b.device)
    q0 = None
    if x0 is not None and bool(x0.any()):
        q0 = torch.empty_like(x)
        matvec_dev(x, q0)
    _lib.check(lib.vt_cg_init(b.data_ptr(), dev.ptr(q0), r.data_ptr(), p.data_ptr(), D, scal.data_ptr(), st),
               "vt_cg_init")
    h = scal.cpu()
    rnorm, bnorm = float(h[1]), float(h[4])
    if rnorm <= tol:                     # SciPy's legacy early exit (_get_atol: resid <= tol)
        return x, 0, 0
    atol = tol if bnorm == 0 else tol * bnorm
    for it in range(maxiter):
        if rnorm < atol:
            return x, 0, it
        matvec_dev(p, q)
        _lib.check(lib.vt_cg_update(p.data_ptr(), q.data_ptr(), x.data_ptr(), r.data_ptr(), D,
                                    scal.data_ptr(), st), "vt_cg_update")
        rnorm = float(scal[1].item())    # the iteration's only host sync (8 bytes)
    return x, maxiter, maxiter


def make_cg_solver(mat_times_vec, dim, cg_opts={}):
    dev.require_gpu()
    opts = dict(cg_opts)
    tol = opts.pop('tol', opts.pop('rtol', 1e-5))
    maxiter = opts.pop('maxiter', None)
    x0 = opts.pop('x0', None)
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

    def solve(v):
        b = dev.to_device(v).contiguous().reshape(-1)
        if b.shape[0] != dim:
            raise ValueError('shapes of A ({0}, {0}) and b {1} are incompatible'.format(dim, tuple(b.shape)))
        x0d = None if x0 is None else dev.to_device(x0).contiguous()
        x, info, _ = cg_solve(matvec_dev, b, x0=x0d, tol=tol, maxiter=maxiter)
        if info != 0:
            warnings.warn('CG exited with error code {}'.format(info))
        return dev.like_input(x, v)
    return solve
