"""ORACLE (test infrastructure, never a product path): the first two directional derivatives
of the optimum in the data weights, restating what
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


def input_derivs(family, X, y, theta, w, dw, prior=None, scale=1.0, offset=None, max_order=2,
                 max_input_order=None):
    """[d theta/dw . dw,  d^2 theta/dw^2 [dw, dw], ...] at (theta, w) (reference: evaluate_input_derivs,
    sensitivity_lib.py:1273-1285).  ``max_input_order``: the reference drops every term whose number of
    partial derivatives w.r.t. the input (``total_eta_order``) exceeds it (:1247-1249); a term
    ``w l_{m+1} (...) x`` is an m-th input partial of G, a term ``dw l_{m+1} (...) x`` an m-th input and
    first hyper partial (l_k = k-th link derivative)."""
    if max_order < 1 or (max_order > 3 and family != "gaussian"):
        raise ValueError("oracle covers orders 1 to 3 (any order for the gaussian family)")
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
