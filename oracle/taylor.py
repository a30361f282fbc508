"""ORACLE (test infrastructure, never a product path): the directional derivatives
of the optimum in the data weights (closed forms for orders 1-3, power-series composition for any order), restating what
``ParametricSensitivityTaylorExpansion`` (``/root/reference/vittles/sensitivity_lib.py:1021-1363``)
evaluates for the registered weighted-GLM objective.

The reference Taylor-expands the estimating equation ``G(theta, w) = grad_theta f`` with nested
``autograd.make_jvp`` calls and solves, order by order (``_evaluate_dkinput_dhyperk``, :1209-1260),

    d^k theta = -H^{-1} ( sum of all order-k terms that do not contain d^k theta itself ).

For ``f = scale * sum_n w_n ell(x_n.theta + off_n, y_n) + prior`` (linear in w):
    order 1 (:_get_taylor_base_terms, 1007-1018):  G_w[dw]               = X^T (scale ell' .* dw)
    order 2 (DerivativeTerm.differentiate, 575-640): G_tt[d1, d1] + 2 G_tw[d1, dw]
            = X^T ( scale * (w ell''' t^2 + 2 dw ell'' t) ),   t = X d1.
``evaluate_taylor_series`` (:1303-1342) then sums ``theta0 + d1 + d2 / 2``.
"""
import numpy as np
import scipy.linalg

from . import glm


def link3(family, z):
    """ell''(z), ell'''(z)."""
    if family == "logistic":
        p = glm._sigmoid(z)
        l2 = p * (1.0 - p)
        return l2, l2 * (1.0 - 2.0 * p)
    if family == "poisson":
        ez = np.exp(z)
        return ez, ez
    if family == "gaussian":
        return np.ones_like(z), np.zeros_like(z)
    raise ValueError(family)


def link4(family, z):
    """Fourth link derivative d^4 ell / dz^4."""
    l2, _ = link3(family, z)
    if family == "logistic":
        return l2 * (1.0 - 6.0 * l2)
    if family == "poisson":
        return l2
    return np.zeros_like(z)


def _stirling2(n, k):
    """Stirling numbers of the second kind (exact integers)."""
    if n == k:
        return 1
    if k == 0 or k > n:
        return 0
    return k * _stirling2(n - 1, k) + _stirling2(n - 1, k - 1)


def link_derivs(family, z, kmax):
    """{m: d^m ell / dz^m} for m = 2..kmax.  Logistic: ell' = sigma(z) - y and the closed form
    sigma^(n) = sum_{j=1..n+1} (-1)^(j-1) (j-1)! S(n+1, j) sigma^j  (S: Stirling numbers of the second kind)."""
    from math import factorial
    out = {}
    if family == "logistic":
        p = glm._sigmoid(z)
        for m in range(2, kmax + 1):
            n = m - 1
            out[m] = sum((-1.0) ** (j - 1) * factorial(j - 1) * _stirling2(n + 1, j) * p ** j for j in range(1, n + 2))
    elif family == "poisson":
        ez = np.exp(z)
        for m in range(2, kmax + 1):
            out[m] = ez
    elif family == "gaussian":
        for m in range(2, kmax + 1):
            out[m] = np.ones_like(z) if m == 2 else np.zeros_like(z)
    else:
        raise ValueError(family)
    return out


def rhs_coeff(family, z, w, dw, zetas, k, max_input_order=None, scale=1.0):
    """Per-row coefficient c of the order-k equation  H d_k = -X^T c  by POWER-SERIES composition: with
    delta(eps) = sum_{j<k} zeta_j eps^j / j!  (zeta_j = X d_j; the unknown zeta_k is left out -- it is the H d_k term),
    c = scale * k! * [eps^k] (w + eps dw) * sum_{m=1..mio} ell^(m+1)(z) delta(eps)^m / m!.
    The m-th power of delta carries m input partial derivatives of the estimating equation, which is what the
    reference's ``max_input_order`` truncates (:1247-1249)."""
    from math import factorial
    N = len(z)
    mio = k if max_input_order is None else min(int(max_input_order), k)
    ld = link_derivs(family, z, k + 1)
    delta = np.zeros((N, k + 1))
    for j, zj in enumerate(zetas[:k - 1], start=1):
        delta[:, j] = zj / factorial(j)
    comp = np.zeros((N, k + 1))
    pw = np.zeros((N, k + 1))
    pw[:, 0] = 1.0
    for m in range(1, mio + 1):
        nxt = np.zeros_like(pw)
        for a in range(k + 1):              # truncated product pw * delta
            for b in range(1, k + 1 - a):
                nxt[:, a + b] += pw[:, a] * delta[:, b]
        pw = nxt
        comp += (ld[m + 1] / factorial(m))[:, None] * pw
    total = w[:, None] * comp
    total[:, 1:] += dw[:, None] * comp[:, :-1]
    return scale * factorial(k) * total[:, k]


def input_derivs_general(family, X, y, theta, w, dw, prior=None, scale=1.0, offset=None, max_order=4,
                         max_input_order=None):
    """``input_derivs`` for ANY order through ``rhs_coeff`` (orders 1..3 agree with the closed forms below)."""
    z = glm.linear_predictor(X, theta, offset)
    _, l1, _ = glm.link(family, z, y)
    chol = scipy.linalg.cho_factor(glm.hessian(family, X, y, theta, w, prior, scale, offset))
    out = [-scipy.linalg.cho_solve(chol, X.T @ (scale * l1 * dw))]
    for k in range(2, max_order + 1):
        c = rhs_coeff(family, z, w, dw, [X @ d for d in out], k, max_input_order, scale)
        out.append(-scipy.linalg.cho_solve(chol, X.T @ c))
    return out


def input_derivs(family, X, y, theta, w, dw, prior=None, scale=1.0, offset=None, max_order=2,
                 max_input_order=None):
    """[d theta/dw . dw,  d^2 theta/dw^2 [dw, dw], ...] at (theta, w) (reference: evaluate_input_derivs,
    sensitivity_lib.py:1273-1285).  ``max_input_order``: the reference drops every term whose number of
    partial derivatives w.r.t. the input (``total_eta_order``) exceeds it (:1247-1249); a term
    ``w l_{m+1} (...) x`` is an m-th input partial of G, a term ``dw l_{m+1} (...) x`` an m-th input and
    first hyper partial (l_k = k-th link derivative)."""
    if max_order < 1:
        raise ValueError("max_order must be at least one")
    if max_order > 3 and family != "gaussian":
        return input_derivs_general(family, X, y, theta, w, dw, prior, scale, offset, max_order, max_input_order)
    z = glm.linear_predictor(X, theta, offset)
    _, l1, _ = glm.link(family, z, y)
    l2, l3 = link3(family, z)
    chol = scipy.linalg.cho_factor(glm.hessian(family, X, y, theta, w, prior, scale, offset))
    d1 = -scipy.linalg.cho_solve(chol, X.T @ (scale * l1 * dw))
    out = [d1]
    mio = np.inf if max_input_order is None else max_input_order
    if max_order >= 2:
        t = X @ d1
        rhs = X.T @ (scale * ((w * l3 * t * t if mio >= 2 else 0.0) + 2.0 * dw * l2 * t))
        out.append(-scipy.linalg.cho_solve(chol, rhs))
    if max_order >= 3 and family != "gaussian":
        # third total derivative of (w + t dw) * dl(z(t)), z(t) = z + t z1 + t^2/2 z2 + t^3/6 z3, without the
        # w l2 z3 term (that one is H d3):  w (l4 z1^3 + 3 l3 z1 z2) + 3 dw (l3 z1^2 + l2 z2)
        z1, z2 = X @ out[0], X @ out[1]
        l4 = link4(family, z)
        rhs = X.T @ (scale * ((w * l4 * z1 ** 3 if mio >= 3 else 0.0) +
                              (3.0 * w * l3 * z1 * z2 + 3.0 * dw * l3 * z1 ** 2 if mio >= 2 else 0.0) +
                              3.0 * dw * l2 * z2))
        out.append(-scipy.linalg.cho_solve(chol, rhs))
        return out
    for k in range(3, max_order + 1):
        # gaussian: G bilinear in (theta, w)  =>  H d_k + k G_tw[d_{k-1}, dw] = 0
        rhs = k * (X.T @ (scale * dw * (X @ out[-1])))
        out.append(-scipy.linalg.cho_solve(chol, rhs))
    return out


def taylor_series(family, X, y, theta, w, w_new, prior=None, scale=1.0, offset=None, max_order=2,
                  add_offset=True):
    """``evaluate_taylor_series`` (sensitivity_lib.py:1303-1342)."""
    ds = input_derivs(family, X, y, theta, w, w_new - w, prior, scale, offset, max_order)
    total = theta.copy() if add_offset else np.zeros_like(theta)
    fact = 1.0
    for k, d in enumerate(ds, start=1):
        fact *= k
        total = total + d / fact
    return total
