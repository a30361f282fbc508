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

from . import _lib, device as dev, sharding


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


# The tile kernels (and the LDS-staged large-D ones) move 16-byte lanes straight from X into LDS: they need an even
# leading dimension and a 16-byte aligned base, and every other layout runs the register-streaming kernels
# (D = 97 contiguous: 0.375 ms against 0.255 ms at N = 10^6).  A matrix that does not have that layout is therefore
# re-laid ONCE at construction, rows padded to an even stride, when the copy fits comfortably in free HBM.
# Tests set this to False to keep the unaligned kernels covered.
RELAYOUT_UNALIGNED_ROWS = True


def _aligned_rows(X):
    N, D = X.shape
    # narrow designs (D <= 80) run the row-link kernel, which takes any 8-byte aligned rows, and are HBM-bound:
    # padded rows would only add traffic (D = 15, N = 10^6: solve 0.040 ms dense, 0.063 ms padded)
    if X.dtype == torch.float32 and D <= 128:
        return X                      # the fp32 mode's resident kernels take any 4-byte aligned rows
    if not RELAYOUT_UNALIGNED_ROWS or N == 0 or D <= 80:
        return X
    per16 = 16 // X.element_size()    # elements per 16-byte LDS-DMA lane (float rows of the fp32 mode at D > 128: 4)
    if X.stride(0) % per16 == 0 and X.data_ptr() % 16 == 0:
        return X
    Dp = (D + per16 - 1) // per16 * per16
    free, _ = torch.cuda.mem_get_info(X.device)
    if 2 * N * Dp * X.element_size() > free:
        return X
    buf = torch.empty((N, Dp), dtype=X.dtype, device=X.device)
    buf[:, :D] = X
    if Dp > D:
        buf[:, D:] = 0.0
    return buf[:, :D]


class WeightedGLMObjective(RegisteredObjective):
    r"""f(theta, w) = scale * sum_n w_n ell(x_n . theta + offset_n, y_n) + 1/2 sum_j prior_j theta_j^2

    family: 'logistic' | 'poisson' | 'gaussian'.  ``X`` (N, D), ``y`` (N,) may be numpy
    or torch; they are moved to the GPU once and stay resident.  ``process_group``:
    a torch.distributed group whose ranks each hold a contiguous ROW shard of
    (X, y, w); partial Hessians / gradients are all-reduced (RCCL), the prior is
    added once, sensitivities stay sharded like X.

    ``dtype=torch.float32`` selects the fp32 STORAGE mode (BASELINE.json north_star: optional, 1e-4
    against the fp64 oracle, never the headline): X is kept -- and the sensitivity matrix returned -- in
    float, half the HBM bytes; theta, y, w, r, d, H, g and the Cholesky factor stay double.  Hessian route: D <= 4096 --
    up to 128 the f32-MFMA resident kernels, beyond the tiled kernels on float panels with fp64 products and a
    slab-wise fp64 solve (configs[2]'s D = 1000 with X and S at half the footprint).  Row operators (link, gradient,
    value, HVP and the CG route, A v, A^T u): D <= 20448, rows are re-laid on a 16-byte pitch here (configs[3]'s
    D = 20000 on a float X)."""

    def __init__(self, X, y, family="logistic", prior=None, scale=1.0, offset=None,
                 device=None, process_group=None, dtype=torch.float64):
        dev.require_gpu()
        if family not in _lib.FAMILY:
            raise ValueError("family must be one of %s" % (tuple(_lib.FAMILY),))
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("dtype must be torch.float64 or torch.float32")
        self.family = family
        self.dtype = dtype
        self.X = dev.to_device(X, device, dtype=dtype)
        if self.X.ndim != 2:
            raise ValueError("X must be (N, D)")
        if self.X.shape[0] > 0 and self.X.stride(1) != 1:
            self.X = self.X.contiguous()
        self.X = _aligned_rows(self.X)
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
        self.stage_hook = None      # optional callable(name): 'allreduce_begin' / 'allreduce_end' (bench timers)

    def invalidate_cache(self):
        """Forget the cached fused pass.  The cache key is (data_ptr, torch version counter) of
        (theta, w): a tensor rewritten behind torch's back (a raw kernel, a DLPack alias) keeps
        its key -- call this after such an update."""
        self._cache_key = None
        self._cache = None

    # ---- distributed plumbing ------------------------------------------------
    def _rank(self):
        return sharding.group_rank(self.process_group)

    def _all_reduce(self, t):
        return sharding.all_reduce_sum(t, self.process_group)

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
        f32 = self.dtype == torch.float32
        nbytes = (lib.vt_glm_hessian_f32_workspace_bytes if f32 else lib.vt_glm_hessian_workspace_bytes)(
            self.N, self.D)
        if nbytes == 0:
            raise _lib.VittlesHipError("vt_glm_hessian%s does not support D = %d"
                                       % ("_f32 (fp32 mode: D <= 4096 for the Hessian route; the CG route takes D <= 20448)" if f32 else "", self.D))
        # scratch per stream: two streams driving one objective never share it
        skey = dev.stream_ptr()
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(skey)
        if ws is None or ws.numel() * 8 < nbytes:
            ws = self._ws[skey] = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=self.device)
        # H and g are views of ONE buffer, so that the row-sharded path all-reduces a single
        # message of D*D + D doubles in place (no gather / scatter copies around the collective)
        # (g starts on a 256-byte boundary; the few padding doubles in between ride along)
        goff = (self.D * self.D + 31) // 32 * 32
        Hg = torch.empty(goff + self.D, dtype=torch.float64, device=self.device)
        H = Hg[:self.D * self.D].view(self.D, self.D)
        g = Hg[goff:]
        r = torch.empty(self.N, dtype=torch.float64, device=self.device)
        d = torch.empty(self.N, dtype=torch.float64, device=self.device)
        prior = self.prior if self._rank() == 0 else None
        _lib.check(_lib.fn(lib, "vt_glm_hessian", self.X)(
            _lib.FAMILY[self.family], self.X.data_ptr(), dev.ld(self.X), theta.data_ptr(),
            self.y.data_ptr(), w.data_ptr(), dev.ptr(self.offset), self.scale, dev.ptr(prior),
            self.N, self.D, H.data_ptr(), self.D, g.data_ptr(), r.data_ptr(), d.data_ptr(),
            ws.data_ptr(), ws.numel() * 8, dev.stream_ptr()), "vt_glm_hessian")
        if self.stage_hook is not None:
            self.stage_hook("allreduce_begin")
        self._all_reduce(Hg)          # one RCCL message: D*D + D doubles
        if self.stage_hook is not None:
            self.stage_hook("allreduce_end")
        out = dict(H=H, g=g, r=r, d=d, theta=theta, w=w)
        self._cache_key, self._cache = key, out
        return out

    def grad(self, opt, hyper):
        if self.dtype == torch.float32:
            # fp32 mode: the fused kernel's gradient is summed in f32 per wave (~1e-6 of sum |terms|), too coarse
            # for an optimiser's stopping test; the row kernel carries the sum in fp64 (one more pass over X)
            from . import rows
            g, _, _ = rows.glm_grad(self, opt, hyper, prior=self.prior if self._rank() == 0 else None)
            return self._all_reduce(g)
        return self.fused(opt, hyper)["g"]

    def hessian(self, opt, hyper):
        return self.fused(opt, hyper)["H"]

    def cross_hessian(self, opt, hyper):
        return LazyCrossHessian(self.X, self.fused(opt, hyper)["r"])

    def value(self, opt, hyper):
        from . import rows
        return rows.glm_value(self, opt, hyper)


class HierGaussMFVIObjective(RegisteredObjective):
    r"""Hierarchical mean-field Gaussian VI objective (KL(q || posterior) up to a constant):

        y_gi ~ N(x_gi . theta_g, 1/tau_y),  theta_g ~ N(mu, I/tau_th),  mu ~ N(0, I/tau_mu)
        q(theta_g) = N(m_g, diag v_g),  q(mu) = N(m_mu, diag v_mu)

    ``opt_par`` layout: [m_0, v_0, ..., m_{G-1}, v_{G-1}, m_mu, v_mu] (each piece k long),
    d = 2k(G+1); the Hessian is block diagonal (G blocks of 2k) plus a 2k-wide arrowhead
    border for the globals.  ``hyper_par`` = per-row data weights (G*ng,), default 1.
    It is a callable ``f(opt_par[, weights])`` like the reference expects, and carries its
    gradient and HVP as HIP kernels (the reference would use autograd)."""

    def __init__(self, X, y, G, ng, k, tau_y=1.0, tau_th=1.0, tau_mu=1.0, device=None, process_group=None):
        """``process_group``: the GROUPS are sharded over the ranks of that torch.distributed group (one process
        per GPU): this rank holds ``G`` of them (``X``, ``y`` of its groups only) and its
        ``opt_par`` is [its groups' (m_g, v_g) ..., m_mu, v_mu] with the globals replicated.  Gradients / HVPs
        return this rank's local rows plus the ALL-REDUCED global rows (2k doubles), the value is all-reduced;
        mu's prior and entropy are contributed by rank 0 only."""
        dev.require_gpu()
        self.G, self.ng, self.k = int(G), int(ng), int(k)
        self.X = dev.to_device(X, device).reshape(self.G * self.ng, self.k).contiguous()
        self.device = self.X.device
        self.y = dev.to_device(y, self.device).reshape(-1).contiguous()
        if self.y.shape[0] != self.G * self.ng:
            raise ValueError("y must have G*ng entries")
        self.tau = (float(tau_y), float(tau_th), float(tau_mu))
        self.d = 2 * self.k * (self.G + 1)
        self.n_global = 2 * self.k                  # trailing entries of opt_par shared by all ranks
        self.process_group = process_group
        self._global_terms = 1 if sharding.group_rank(process_group) == 0 else 0
        self._ones = torch.ones(self.G * self.ng, dtype=torch.float64, device=self.device)
        lib = _lib.load()
        self._ws = torch.empty(lib.vt_vi_workspace_bytes(self.G, self.k) // 8 + 1, dtype=torch.float64,
                               device=self.device)

    def _reduce_global_rows(self, out):
        """Sum the partial global rows (last 2k entries) over the ranks of a group-sharded objective."""
        if self.process_group is not None:
            sharding.all_reduce_sum(out[self.d - self.n_global:], self.process_group)
        return out

    def sparsity_array(self):
        return np.arange(2 * self.k * self.G).reshape(self.G, 2 * self.k)

    def global_inds(self):
        return np.arange(2 * self.k * self.G, self.d)

    def _w(self, hyper):
        return self._ones if hyper is None else dev.to_device(hyper, self.device).contiguous()

    def _check(self, opt):
        opt = dev.to_device(opt, self.device).contiguous()
        if opt.shape != (self.d,):
            raise ValueError("opt_par must have shape (%d,)" % self.d)
        return opt

    def __call__(self, opt_par, hyper_par=None):
        return self.value(opt_par, hyper_par)

    def value(self, opt, hyper=None):
        lib = dev.require_gpu()
        opt = self._check(opt)
        val = torch.empty(1, dtype=torch.float64, device=self.device)
        _lib.check(lib.vt_vi_value(self.X.data_ptr(), self.y.data_ptr(), self._w(hyper).data_ptr(), opt.data_ptr(),
                                   *self.tau, self.G, self.ng, self.k, self._global_terms, val.data_ptr(),
                                   self._ws.data_ptr(), self._ws.numel() * 8, dev.stream_ptr()), "vt_vi_value")
        sharding.all_reduce_sum(val, self.process_group)
        return float(val.item())

    def grad(self, opt, hyper=None):
        lib = dev.require_gpu()
        opt = self._check(opt)
        out = torch.empty(self.d, dtype=torch.float64, device=self.device)
        _lib.check(lib.vt_vi_grad(self.X.data_ptr(), self.y.data_ptr(), self._w(hyper).data_ptr(), opt.data_ptr(),
                                  *self.tau, self.G, self.ng, self.k, self._global_terms, out.data_ptr(),
                                  self._ws.data_ptr(), self._ws.numel() * 8, dev.stream_ptr()), "vt_vi_grad")
        return self._reduce_global_rows(out)

    def hvp(self, opt, vec, hyper=None, out=None):
        lib = dev.require_gpu()
        opt = self._check(opt)
        vec = dev.to_device(vec, self.device).contiguous()
        if out is None:
            out = torch.empty(self.d, dtype=torch.float64, device=self.device)
        _lib.check(lib.vt_vi_hvp(self.X.data_ptr(), self._w(hyper).data_ptr(), opt.data_ptr(), vec.data_ptr(),
                                 *self.tau, self.G, self.ng, self.k, self._global_terms, out.data_ptr(),
                                 self._ws.data_ptr(), self._ws.numel() * 8, dev.stream_ptr()), "vt_vi_hvp")
        return self._reduce_global_rows(out)

    def hessian(self, opt, hyper=None):
        """Dense (d, d) Hessian by d HVPs -- small problems only (tests, validation)."""
        if self.d > 4096:
            raise _lib.VittlesHipError("dense Hessian of a d=%d VI objective: use SparseBlockHessian" % self.d)
        H = torch.empty((self.d, self.d), dtype=torch.float64, device=self.device)
        e = torch.zeros(self.d, dtype=torch.float64, device=self.device)
        for j in range(self.d):
            e[j] = 1.0
            self.hvp(opt, e, hyper, out=H[j])
            e[j] = 0.0
        return H   # symmetric: rows == columns
