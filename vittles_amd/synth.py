"""Counter-based synthetic data (SURVEY.md section 8d "Synthetic inputs"): Philox-4x32-10, key = seed,
counter = (row, column pair, stream).  Device side: ``vt_synth_normal`` / ``vt_synth_response`` (csrc/vt_synth.hip);
host side: the same generator in NumPy, so that any block of rows of a data set that lives only in HBM can be
regenerated on the host for a spot check (identical random bits; the Box-Muller transform agrees to a few ulps of the
two libms), and so that the row shards of N ranks are slices of ONE global data set whatever N is.

The reference has no generator (its tests seed np.random, vittles/tests/test_sensitivity_lib.py:463-478); this is
set-up code in front of the hot path, not part of it.
"""
import numpy as np
import torch

from . import _lib, device as dev, rows

STREAM_X, STREAM_Y, STREAM_THETA = 0, 1, 2
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, seed):
    """Philox-4x32-10 on uint64 arrays holding 32-bit words; returns the four output words."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _M0 * c0, _M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0), p1 & _MASK,
                          (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1), p0 & _MASK)
        k0, k1 = (k0 + _W0) & 0xFFFFFFFF, (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _u01(hi, lo):
    m = ((hi << np.uint64(32)) | lo) >> np.uint64(11)
    return (m.astype(np.float64) + 0.5) * 2.0 ** -53


def _sincospi(x):
    """sin(pi x), cos(pi x) for x in (0, 2) with the argument reduced exactly (as the device's sincospi does)."""
    k = np.rint(2.0 * x)
    r = x - 0.5 * k                       # exact, |r| <= 1/4
    s, c = np.sin(np.pi * r), np.cos(np.pi * r)
    k = k.astype(np.int64) & 3
    sn = np.where(k == 0, s, np.where(k == 1, c, np.where(k == 2, -s, -c)))
    cs = np.where(k == 0, c, np.where(k == 1, -s, np.where(k == 2, -c, s)))
    return sn, cs


def normal_rows_host(seed, row0, nrows, D, stream=STREAM_X, row_index=None):
    """Rows ``row0 .. row0 + nrows - 1`` (or the global rows ``row_index``) of the N(0,1) matrix, on the host."""
    rws = np.arange(row0, row0 + nrows, dtype=np.uint64) if row_index is None else np.asarray(row_index, dtype=np.uint64)
    P = (D + 1) // 2
    rr = np.repeat(rws, P)
    pp = np.tile(np.arange(P, dtype=np.uint64), len(rws))
    x0, x1, x2, x3 = philox4x32_10(rr & _MASK, rr >> np.uint64(32), pp, np.full_like(pp, stream), seed)
    u1, u2 = _u01(x0, x1), _u01(x2, x3)
    rad = np.sqrt(-2.0 * np.log(u1))
    sn, cs = _sincospi(2.0 * u2)
    out = np.empty((len(rws), 2 * P))
    out[:, 0::2] = (rad * cs).reshape(len(rws), P)
    out[:, 1::2] = (rad * sn).reshape(len(rws), P)
    return np.ascontiguousarray(out[:, :D])


def response_host(seed, z, family, row0=0, row_index=None):
    z = np.asarray(z, dtype=np.float64)
    rws = np.arange(row0, row0 + len(z), dtype=np.uint64) if row_index is None else np.asarray(row_index, dtype=np.uint64)
    zero = np.zeros_like(rws)
    x0, x1, x2, x3 = philox4x32_10(rws & _MASK, rws >> np.uint64(32), zero, zero + np.uint64(STREAM_Y), seed)
    u = _u01(x0, x1)
    if family == "logistic":
        return (u < 1.0 / (1.0 + np.exp(-z))).astype(np.float64)
    if family == "poisson":
        lam = np.exp(z)
        pk = np.exp(-lam)
        cdf = pk.copy()
        k = np.zeros(len(z))
        live = cdf < u
        it = 0
        while live.any() and it < 100000:
            it += 1
            k[live] += 1
            pk[live] *= lam[live] / k[live]
            cdf[live] += pk[live]
            live = live & (cdf < u)
        return k
    _, cs = _sincospi(2.0 * _u01(x2, x3))
    return z + np.sqrt(-2.0 * np.log(u)) * cs


def theta_star_host(seed, D, scale=None):
    """theta*_j ~ N(0, 1/D) (logit standard deviation ~ 1), identical on every rank."""
    t = normal_rows_host(seed, 0, 1, D, stream=STREAM_THETA)[0]
    return t * (1.0 / np.sqrt(D) if scale is None else scale)


def normal_rows(seed, row0, nrows, D, device=None, dtype=torch.float64, stream=STREAM_X, out=None):
    """The same rows on the device (``vt_synth_normal``)."""
    lib = dev.require_gpu()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if out is None:
        out = torch.empty((nrows, D), dtype=dtype, device=device)
    fn = lib.vt_synth_normal_f32 if out.dtype == torch.float32 else lib.vt_synth_normal
    _lib.check(fn(int(seed), int(stream), int(row0), int(nrows), int(D), out.data_ptr(), dev.ld(out),
                  dev.stream_ptr()), "vt_synth_normal")
    return out


def glm_problem(N, D, family="logistic", seed=0, row0=0, device=None, dtype=torch.float64, logit_scale=None):
    """Rows ``row0 .. row0 + N - 1`` of the global synthetic GLM data set of SURVEY.md section 8d, generated in HBM:
    ``X[n, j] ~ N(0,1)``, ``theta*_j ~ N(0, 1/D)``, ``y_n ~`` Bernoulli(sigmoid(z)) / Poisson(exp(z)) / N(z, 1) with
    ``z = logit_scale * x_n . theta*`` (default 1, 0.5 for the poisson family).  Returns ``(X, y, theta_star)`` as CUDA
    tensors; ``y`` and ``theta_star`` are float64 whatever ``dtype`` (the storage type of X) is."""
    lib = dev.require_gpu()
    if logit_scale is None:
        logit_scale = 0.5 if family == "poisson" else 1.0
    X = normal_rows(seed, row0, N, D, device=device, dtype=dtype)
    theta = torch.as_tensor(theta_star_host(seed, D), device=X.device)
    z = rows.gemv(X, theta * logit_scale)
    y = torch.empty(N, dtype=torch.float64, device=X.device)
    _lib.check(lib.vt_synth_response(int(seed), int(row0), int(N), _lib.FAMILY[family], z.data_ptr(), y.data_ptr(),
                                     dev.stream_ptr()), "vt_synth_response")
    return X, y, theta


def glm_rows_host(N_total, D, family, seed, row_index, logit_scale=None, dtype=np.float64):
    """Host regeneration of arbitrary global rows of ``glm_problem``'s data set: ``(X_rows, y_rows)``.
    ``dtype=np.float32`` reproduces the fp32 storage mode's X (rounded, then used in fp64 for z)."""
    if logit_scale is None:
        logit_scale = 0.5 if family == "poisson" else 1.0
    row_index = np.asarray(row_index)
    if row_index.size and (row_index.min() < 0 or row_index.max() >= N_total):
        raise ValueError("row_index out of range")
    X = normal_rows_host(seed, 0, 0, D, row_index=row_index).astype(dtype).astype(np.float64)
    z = X @ (theta_star_host(seed, D) * logit_scale)
    return X, response_host(seed, z, family, row_index=row_index)
