"""Host wrappers over the row-streaming kernels (vt_rows.hip): K0, K3, K6, K9, value."""
import torch

from . import _lib, device as dev

_ws_cache = {}


def _grow_only(cache, device, nbytes):
    """One buffer per (device, STREAM), reallocated only when a call needs more: two streams of one process never
    share scratch (kernels on different streams may overlap), and a sweep over D does not pin a buffer per size."""
    key = (device.index, dev.stream_ptr())
    ws = cache.get(key)
    if ws is None or ws.numel() * 8 < nbytes:
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=device)
        cache[key] = ws
    return ws


def _workspace(D, device):
    """Scratch of the row-streaming kernels (sized for the CU count of ``device``, not of the current device)."""
    lib = dev.require_gpu()
    with torch.cuda.device(device):
        nbytes = lib.vt_rows_workspace_bytes(int(D))
    return _grow_only(_ws_cache, device, nbytes)


def dot(a, b, c=None):
    """sum_i a_i b_i (c_i) of CUDA fp64 vectors as a 1-element CUDA tensor (fixed-order, one launch)."""
    lib = dev.require_gpu()
    a, b = a.contiguous().reshape(-1), b.contiguous().reshape(-1)
    if a.shape != b.shape or (c is not None and c.shape != a.shape):
        raise ValueError("dot: shapes differ")
    if c is not None:
        c = c.contiguous().reshape(-1)
    out = torch.empty(1, dtype=torch.float64, device=a.device)
    _lib.check(lib.vt_dot3(a.data_ptr(), b.data_ptr(), dev.ptr(c), a.shape[0], out.data_ptr(), dev.stream_ptr()),
               "vt_dot3")
    return out


def norm(a):
    """2-norm of a CUDA fp64 vector as a Python float."""
    return float(dot(a, a).sqrt().item())


def _rowmajor(A):
    return A if (A.stride(1) == 1 or A.shape[0] == 0) else A.contiguous()


def _fp64_only(A, what):
    if A.dtype != torch.float64:
        raise _lib.VittlesHipError(
            "%s is not part of the fp32 storage mode (it covers the Hessian, the sensitivity solve, predictions, gradient, HVP, "
            "value, A v and A^T u): build the objective with dtype=torch.float64" % what)


def gemv_t(A, u):
    """out (D,) = A^T u for a row-major (N, D) CUDA tensor A (K9)."""
    lib = dev.require_gpu()
    A = _rowmajor(A)
    N, D = A.shape
    u = u.contiguous()
    out = torch.empty(D, dtype=torch.float64, device=A.device)
    ws = _workspace(D, A.device)
    _lib.check(_lib.fn(lib, "vt_gemv_t", A)(A.data_ptr(), dev.ld(A), u.data_ptr(), N, D, out.data_ptr(),
                                            ws.data_ptr(), ws.numel() * 8, dev.stream_ptr()), "vt_gemv_t")
    return out


def gemv(A, v):
    """out (N,) = A v for a row-major (N, D) CUDA tensor A."""
    lib = dev.require_gpu()
    A = _rowmajor(A)
    N, D = A.shape
    v = v.contiguous()
    out = torch.empty(N, dtype=torch.float64, device=A.device)
    ws = _workspace(D, A.device)
    _lib.check(_lib.fn(lib, "vt_gemv", A)(A.data_ptr(), dev.ld(A), v.data_ptr(), N, D, out.data_ptr(),
                                          ws.data_ptr(), ws.numel() * 8, dev.stream_ptr()), "vt_gemv")
    return out


def gemv_t2(A, u1, u2):
    """out (D,) = A^T (u1 .* u2) for a row-major (N, D) CUDA tensor A."""
    lib = dev.require_gpu()
    A = _rowmajor(A)
    N, D = A.shape
    out = torch.empty(D, dtype=torch.float64, device=A.device)
    ws = _workspace(D, A.device)
    _lib.check(_lib.fn(lib, "vt_gemv_t2", A)(A.data_ptr(), dev.ld(A), u1.contiguous().data_ptr(),
                                             u2.contiguous().data_ptr(), N, D, out.data_ptr(), ws.data_ptr(),
                                             ws.numel() * 8, dev.stream_ptr()), "vt_gemv_t2")
    return out


def glm_taylor2_rhs(obj, z, w, dw, u):
    """Second-order data-weight right-hand side of a WeightedGLMObjective (one pass over X):
    scale * sum_n [w_n ell'''(z_n) (x_n.u)^2 + 2 dw_n ell''(z_n) (x_n.u)] x_n."""
    lib = dev.require_gpu()
    _fp64_only(obj.X, "the Taylor expansion")
    out = torch.empty(obj.D, dtype=torch.float64, device=obj.device)
    ws = _workspace(obj.D, obj.device)
    _lib.check(lib.vt_glm_taylor2_rhs(_lib.FAMILY[obj.family], obj.X.data_ptr(), dev.ld(obj.X), z.data_ptr(),
                                      w.contiguous().data_ptr(), dw.contiguous().data_ptr(), obj.scale,
                                      u.contiguous().data_ptr(), obj.N, obj.D, out.data_ptr(), ws.data_ptr(),
                                      ws.numel() * 8, dev.stream_ptr()), "vt_glm_taylor2_rhs")
    return out


def glm_taylor3_rhs(obj, z, w, dw, d1, d2=None, zeta2=None):
    """Third-order data-weight right-hand side of a WeightedGLMObjective: one pass for zeta2 = X d2
    (unless ``zeta2`` is given), one fused pass for the sum (see include/vittles_hip.h)."""
    lib = dev.require_gpu()
    _fp64_only(obj.X, "the Taylor expansion")
    if zeta2 is None:
        zeta2 = gemv(obj.X, d2)
    out = torch.empty(obj.D, dtype=torch.float64, device=obj.device)
    ws = _workspace(obj.D, obj.device)
    _lib.check(lib.vt_glm_taylor3_rhs(_lib.FAMILY[obj.family], obj.X.data_ptr(), dev.ld(obj.X), z.data_ptr(),
                                      w.contiguous().data_ptr(), dw.contiguous().data_ptr(), zeta2.data_ptr(),
                                      obj.scale, d1.contiguous().data_ptr(), obj.N, obj.D, out.data_ptr(),
                                      ws.data_ptr(), ws.numel() * 8, dev.stream_ptr()), "vt_glm_taylor3_rhs")
    return out


TAYLOR_MAX_ORDER = 8      # VT_TAYLOR_MAX_ORDER of include/vittles_hip.h


def glm_taylor_rhs(obj, z, w, dw, derivs, k, max_input_order=None, zetas=None):
    """Order-``k`` data-weight right-hand side of a WeightedGLMObjective for ANY 2 <= k <= TAYLOR_MAX_ORDER:
    ``X^T c`` with the per-row Faa di Bruno coefficient ``c`` of ``vt_glm_taylor_coeff`` (partial Bell polynomials of
    ``zeta_j = X d_j``, j < k).  ``derivs``: the k - 1 lower derivatives of the optimum; ``zetas``: an optional dict
    j -> zeta_j that is filled and reused across orders (one ``gemv`` per derivative).  Two passes over X per order
    (orders 2 and 3 also have the one-pass kernels ``glm_taylor2_rhs`` / ``glm_taylor3_rhs``)."""
    lib = dev.require_gpu()
    _fp64_only(obj.X, "the Taylor expansion")
    if not 2 <= k <= TAYLOR_MAX_ORDER:
        raise NotImplementedError("Taylor orders 2..%d are built (got %d)" % (TAYLOR_MAX_ORDER, k))
    if len(derivs) < k - 1:
        raise ValueError("order %d needs the %d lower derivatives" % (k, k - 1))
    mio = k if max_input_order is None else int(max_input_order)
    zetas = {} if zetas is None else zetas
    Z = torch.empty((k - 1, obj.N), dtype=torch.float64, device=obj.device)
    for j in range(1, k):
        if j not in zetas:
            zetas[j] = gemv(obj.X, derivs[j - 1])
        Z[j - 1].copy_(zetas[j])
    c = torch.empty(obj.N, dtype=torch.float64, device=obj.device)
    _lib.check(lib.vt_glm_taylor_coeff(_lib.FAMILY[obj.family], z.data_ptr(), w.contiguous().data_ptr(),
                                       dw.contiguous().data_ptr(), Z.data_ptr(), obj.N, obj.scale, k, max(mio, 1),
                                       obj.N, c.data_ptr(), dev.stream_ptr()), "vt_glm_taylor_coeff")
    return gemv_t(obj.X, c)


_tn_ws_cache = {}


def gemm_tn(A, B, rowscale=None, alpha=1.0, bias=None, out=None):
    """out (Da, Db) = alpha * A^T diag(rowscale) B + bias[None, :] for row-major CUDA tensors
    A (N, Da), B (N, Db): K9 for many weight vectors at once (fp64 MFMA, split over N)."""
    lib = dev.require_gpu()
    _fp64_only(A, "gemm_tn (first operand)")
    if B.dtype not in (torch.float64, torch.float32):
        raise TypeError("gemm_tn: B must be float64 or float32")
    A, B = _rowmajor(A), _rowmajor(B)
    N, Da = A.shape
    if B.shape[0] != N:
        raise ValueError("A and B must have the same number of rows")
    Db = B.shape[1]
    nbytes = lib.vt_gemm_tn_workspace_bytes(Da, Db)
    if nbytes == 0:
        raise _lib.VittlesHipError("vt_gemm_tn does not support a (%d, %d) product" % (Da, Db))
    key = (A.device.index, dev.stream_ptr(), nbytes)
    ws = _tn_ws_cache.get(key)
    if ws is None:
        ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=A.device)
        for k in [k for k in _tn_ws_cache if k[:2] == key[:2]]:   # one live workspace per (device, stream)
            del _tn_ws_cache[k]
        _tn_ws_cache[key] = ws
    if out is None:
        out = torch.empty((Da, Db), dtype=torch.float64, device=A.device)
    fn = lib.vt_gemm_tn if B.dtype == torch.float64 else lib.vt_gemm_tn_f32b   # float B: the fp32 mode's S
    _lib.check(fn(A.data_ptr(), dev.ld(A), Da, B.data_ptr(), dev.ld(B), Db, dev.ptr(rowscale),
                              N, float(alpha), dev.ptr(bias), out.data_ptr(), dev.ld(out), ws.data_ptr(),
                              ws.numel() * 8, dev.stream_ptr()), "vt_gemm_tn")
    return out


def gemm_tn_tiled(A, B, tile=8192):
    """``A^T B`` for operands of ANY width: vt_gemm_tn takes at most 8192 columns per operand, so wider ones are cut
    into column panels and every (panel of A, panel of B) pair is one vt_gemm_tn call into its slice of the result
    (no library GEMM on the path)."""
    Da, Db = A.shape[1], B.shape[1]
    if Da <= tile and Db <= tile and Da >= 1 and Db >= 1:
        return gemm_tn(A, B)
    out = torch.empty((Da, Db), dtype=torch.float64, device=A.device)
    for i in range(0, Da, tile):
        Ai = A[:, i:i + tile]
        for j in range(0, Db, tile):
            blk = gemm_tn(Ai, B[:, j:j + tile])
            out[i:i + tile, j:j + tile].copy_(blk)
    return out


def glm_link(obj, theta, w, want_z=False):
    """K0: per-row (z,) r, d for a WeightedGLMObjective."""
    lib = dev.require_gpu()
    theta = dev.to_device(theta, obj.device).contiguous()
    w = dev.to_device(w, obj.device).contiguous()
    z = torch.empty(obj.N, dtype=torch.float64, device=obj.device) if want_z else None
    r = torch.empty(obj.N, dtype=torch.float64, device=obj.device)
    d = torch.empty(obj.N, dtype=torch.float64, device=obj.device)
    ws = _workspace(obj.D, obj.device)
    _lib.check(_lib.fn(lib, "vt_glm_link", obj.X)(_lib.FAMILY[obj.family], obj.X.data_ptr(), dev.ld(obj.X), theta.data_ptr(),
                               obj.y.data_ptr(), w.data_ptr(), dev.ptr(obj.offset), obj.scale, obj.N, obj.D,
                               dev.ptr(z), r.data_ptr(), d.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                               dev.stream_ptr()), "vt_glm_link")
    return z, r, d


def glm_grad(obj, theta, w, prior=None):
    """K3: g = X^T (w r) (+ prior theta); also returns r, d."""
    lib = dev.require_gpu()
    theta = dev.to_device(theta, obj.device).contiguous()
    w = dev.to_device(w, obj.device).contiguous()
    g = torch.empty(obj.D, dtype=torch.float64, device=obj.device)
    r = torch.empty(obj.N, dtype=torch.float64, device=obj.device)
    d = torch.empty(obj.N, dtype=torch.float64, device=obj.device)
    ws = _workspace(obj.D, obj.device)
    _lib.check(_lib.fn(lib, "vt_glm_grad", obj.X)(_lib.FAMILY[obj.family], obj.X.data_ptr(), dev.ld(obj.X), theta.data_ptr(),
                               obj.y.data_ptr(), w.data_ptr(), dev.ptr(obj.offset), obj.scale, dev.ptr(prior),
                               obj.N, obj.D, g.data_ptr(), r.data_ptr(), d.data_ptr(), ws.data_ptr(),
                               ws.numel() * 8, dev.stream_ptr()), "vt_glm_grad")
    return g, r, d


def glm_hvp(X, d, v, prior=None, prior_scalar=0.0, out=None):
    """K6: out = X^T (d .* (X v)) + prior .* v."""
    lib = dev.require_gpu()
    N, D = X.shape
    if out is None:
        out = torch.empty(D, dtype=torch.float64, device=X.device)
    ws = _workspace(D, X.device)
    _lib.check(_lib.fn(lib, "vt_glm_hvp", X)(X.data_ptr(), dev.ld(X), d.data_ptr(), dev.ptr(prior), float(prior_scalar),
                              v.data_ptr(), N, D, out.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                              dev.stream_ptr()), "vt_glm_hvp")
    return out


_hvp_multi_ws_cache = {}


def glm_hvp_multi(X, d, V, prior=None, prior_scalar=0.0, out=None):
    """K6 for K vectors at once: out[k, :] = X^T (d .* (X V[k, :])) + prior .* V[k, :] for row-major CUDA fp64
    ``V`` (K, D), K <= 128 -- X is read twice whatever K is (csrc/vt_large.hip: vt_glm_hvp_multi)."""
    lib = dev.require_gpu()
    _fp64_only(X, "glm_hvp_multi")
    N, D = X.shape
    V = _rowmajor(V)
    K = V.shape[0]
    if V.shape[1] != D:
        raise ValueError("V must be (K, %d)" % D)
    if out is None:
        out = torch.empty((K, D), dtype=torch.float64, device=X.device)
    with torch.cuda.device(X.device):
        nbytes = lib.vt_glm_hvp_multi_workspace_bytes(N, D, K)
    if nbytes == 0:
        raise _lib.VittlesHipError("vt_glm_hvp_multi takes 1 <= K <= 128 vectors (got %d) of D <= 20448" % K)
    ws = _grow_only(_hvp_multi_ws_cache, X.device, nbytes)
    _lib.check(lib.vt_glm_hvp_multi(X.data_ptr(), dev.ld(X), d.data_ptr(), dev.ptr(prior), float(prior_scalar),
                                    V.data_ptr(), dev.ld(V), K, N, D, out.data_ptr(), dev.ld(out), ws.data_ptr(),
                                    ws.numel() * 8, dev.stream_ptr()), "vt_glm_hvp_multi")
    return out


def glm_value(obj, theta, w):
    """Objective value f(theta, w) as a Python float."""
    lib = dev.require_gpu()
    theta = dev.to_device(theta, obj.device).contiguous()
    w = dev.to_device(w, obj.device).contiguous()
    val = torch.empty(1, dtype=torch.float64, device=obj.device)
    ws = _workspace(obj.D, obj.device)
    _lib.check(_lib.fn(lib, "vt_glm_value", obj.X)(_lib.FAMILY[obj.family], obj.X.data_ptr(), dev.ld(obj.X), theta.data_ptr(),
                                obj.y.data_ptr(), w.data_ptr(), dev.ptr(obj.offset), obj.scale, obj.N, obj.D,
                                val.data_ptr(), ws.data_ptr(), ws.numel() * 8, dev.stream_ptr()), "vt_glm_value")
    obj._all_reduce(val)
    total = float(val.item())
    if obj.prior is not None:
        total += 0.5 * float(dot(obj.prior, theta, theta).item())
    return total
