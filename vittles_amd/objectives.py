"""Registered objectives: callables ``(opt_par, hyper_par) -> float`` that ALSO expose
their data so the constructors of the drop-in classes can dispatch to HIP kernels
instead of differentiating Python code with autograd.

The reference has no such notion: it takes any Python callable and lets autograd
produce ``grad`` / ``hessian`` / ``jacobian(grad, argnum=1)``
(vittles/sensitivity_lib.py:354-383, 180-181).  Here those three derivatives are
methods of the objective, computed in closed form on the device.
"""
import numpy as np
import torch

from . import _lib, device as dev


class RegisteredObjective:
    """Protocol.  Subclasses implement, for CUDA tensors ``opt`` (D,) and ``hyper`` (M,):
    ``value(opt, hyper) -> float``; ``grad(opt, hyper) -> (D,)``;
    ``hessian(opt, hyper) -> (D, D)``; ``cross_hessian(opt, hyper) -> (D, M)``
    (= d/d hyper of grad, the reference's ``estimating_equation_jac0``)."""

    def __call__(self, opt_par, hyper_par):
        return self.value(dev.to_device(opt_par), dev.to_device(hyper_par))

    def value(self, opt, hyper):
        raise NotImplementedError

    def grad(self, opt, hyper):
        raise NotImplementedError

    def hessian(self, opt, hyper):
        raise NotImplementedError

    def cross_hessian(self, opt, hyper):
        raise NotImplementedError


class LazyCrossHessian:
    """The (D, N) data-weight cross-Hessian ``column n = r_n * x_n`` of a GLM, kept
    as (X, r) so that the N-right-hand-side solve can generate it on the fly
    (K2 fused into K5) instead of materialising D*N doubles."""

    def __init__(self, X, r):
        self.X, self.r = X, r
        self.shape = (X.shape[1], X.shape[0])

    def materialize(self):
        return (self.X * self.r[:, None]).t()


class WeightedGLMObjective(RegisteredObjective):
    r"""f(theta, w) = scale * sum_n w_n ell(x_n . theta + offset_n, y_n) + 1/2 sum_j prior_j theta_j^2

    family: 'logistic' | 'poisson' | 'gaussian'.  ``X`` (N, D), ``y`` (N,) may be numpy
    or torch; they are moved to the GPU once and stay resident.  ``process_group``:
    a torch.distributed group whose ranks each hold a contiguous ROW shard of
    (X, y, w); partial Hessians / gradients are all-reduced (RCCL), the prior is
    added once, sensitivities stay sharded like X."""

    def __init__(self, X, y, family="logistic", prior=None, scale=1.0, offset=None,
                 device=None, process_group=None):
        dev.require_gpu()
        if family not in _lib.FAMILY:
            raise ValueError("family must be one of %s" % (tuple(_lib.FAMILY),))
        self.family = family
        self.X = dev.to_device(X, device)
        if self.X.ndim != 2:
            raise ValueError("X must be (N, D)")
        if self.X.stride(1) != 1:
            self.X = self.X.contiguous()
        self.N, self.D = self.X.shape
        self.device = self.X.device
        self.y = dev.to_device(y, self.device).contiguous()
        if self.y.shape != (self.N,):
            raise ValueError("y must have shape (N,)")
        self.offset = None if offset is None else dev.to_device(offset, self.device).contiguous()
        self.scale = float(scale)
        if prior is None:
            self.prior = None
        else:
            p = dev.to_device(prior, self.device)
            self.prior = (p.expand(self.D) if p.ndim == 0 else p).contiguous()
        self.process_group = process_group
        self._ws = None
        self._cache_key = None
        self._cache = None

    # ---- distributed plumbing ------------------------------------------------
    def _rank(self):
        if self.process_group is None:
            return 0
        import torch.distributed as dist
        return dist.get_rank(self.process_group)

    def _all_reduce(self, t):
        if self.process_group is not None:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.process_group)
        return t

    # ---- fused pass ------------------------------------------------------------
    def fused(self, theta, w):
        """One pass over X: returns dict(H, g, r, d) of CUDA tensors (cached for the
        last (theta, w) pair so grad()/hessian()/cross_hessian() share one pass)."""
        lib = dev.require_gpu()
        theta = dev.to_device(theta, self.device).contiguous()
        w = dev.to_device(w, self.device).contiguous()
        if theta.shape != (self.D,):
            raise ValueError("opt_par must have shape (%d,)" % self.D)
        if w.shape != (self.N,):
            raise ValueError("hyper_par (data weights) must have shape (%d,)" % self.N)
        key = (theta.data_ptr(), w.data_ptr(), theta._version, w._version)
        if self._cache_key == key:
            return self._cache
        nbytes = lib.vt_glm_hessian_workspace_bytes(self.N, self.D)
        if nbytes == 0:
            raise _lib.VittlesHipError("vt_glm_hessian supports D <= %d (got %d)" % (_lib.SMALL_D_MAX, self.D))
        if self._ws is None or self._ws.numel() * 8 < nbytes:
            self._ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=self.device)
        H = torch.empty((self.D, self.D), dtype=torch.float64, device=self.device)
        g = torch.empty(self.D, dtype=torch.float64, device=self.device)
        r = torch.empty(self.N, dtype=torch.float64, device=self.device)
        d = torch.empty(self.N, dtype=torch.float64, device=self.device)
        prior = self.prior if self._rank() == 0 else None
        _lib.check(lib.vt_glm_hessian(
            _lib.FAMILY[self.family], self.X.data_ptr(), self.X.stride(0), theta.data_ptr(),
            self.y.data_ptr(), w.data_ptr(), dev.ptr(self.offset), self.scale, dev.ptr(prior),
            self.N, self.D, H.data_ptr(), self.D, g.data_ptr(), r.data_ptr(), d.data_ptr(),
            self._ws.data_ptr(), self._ws.numel() * 8, dev.stream_ptr()), "vt_glm_hessian")
        self._all_reduce(H)
        self._all_reduce(g)
        out = dict(H=H, g=g, r=r, d=d, theta=theta, w=w)
        self._cache_key, self._cache = key, out
        return out

    def grad(self, opt, hyper):
        return self.fused(opt, hyper)["g"]

    def hessian(self, opt, hyper):
        return self.fused(opt, hyper)["H"]

    def cross_hessian(self, opt, hyper):
        return LazyCrossHessian(self.X, self.fused(opt, hyper)["r"])

    def value(self, opt, hyper):
        from . import rows
        return rows.glm_value(self, opt, hyper)
