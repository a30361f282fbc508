"""ORACLE (test infrastructure, never a product path): closed-form derivatives of
the registered weighted-GLM objective

    f(theta, w) = scale * sum_n w_n * ell(z_n, y_n) + 0.5 * sum_j prior_j * theta_j^2,
    z_n = x_n . theta + offset_n

which is what ``autograd`` would produce for the same callable in the reference:
  * gradient      -> ``autograd.grad(objective_fun, argnum=0)``       sensitivity_lib.py:354
  * Hessian       -> ``autograd.hessian(objective_fun, argnum=0)``    sensitivity_lib.py:382-383
  * cross-Hessian -> ``autograd.jacobian(grad, argnum=1)``  (D, N)    sensitivity_lib.py:180-181, 217-224
  * sens          -> ``-1 * hess_solver(jac0)``                       sensitivity_lib.py:226
"""
import numpy as np
import scipy.linalg

FAMILIES = ("logistic", "poisson", "gaussian")


def _softplus(z):
    return np.logaddexp(0.0, z)


def _sigmoid(z):
    out = np.empty_like(z)
    pos = z >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-z[pos]))
    ez = np.exp(z[~pos])
    out[~pos] = ez / (1.0 + ez)
    return out


def link(family, z, y):
    """Per-row loss ell, d ell/dz, d2 ell/dz2."""
    z = np.asarray(z, dtype=np.float64)
    if family == "logistic":
        p = _sigmoid(z)
        return _softplus(z) - y * z, p - y, p * (1.0 - p)
    if family == "poisson":
        ez = np.exp(z)
        return ez - y * z, ez - y, ez
    if family == "gaussian":
        return 0.5 * (z - y) ** 2, z - y, np.ones_like(z)
    raise ValueError(family)


def _prior_vec(prior, D):
    p = np.zeros(D) if prior is None else np.asarray(prior, dtype=np.float64)
    return np.broadcast_to(p, (D,)).copy()


def linear_predictor(X, theta, offset=None):
    z = X @ theta
    if offset is not None:
        z = z + offset
    return z


def value(family, X, y, theta, w, prior=None, scale=1.0, offset=None):
    ell, _, _ = link(family, linear_predictor(X, theta, offset), y)
    return scale * np.sum(w * ell) + 0.5 * np.sum(_prior_vec(prior, X.shape[1]) * theta ** 2)


def grad(family, X, y, theta, w, prior=None, scale=1.0, offset=None):
    """d f / d theta, shape (D,)."""
    _, l1, _ = link(family, linear_predictor(X, theta, offset), y)
    return scale * (X.T @ (w * l1)) + _prior_vec(prior, X.shape[1]) * theta


def hessian(family, X, y, theta, w, prior=None, scale=1.0, offset=None):
    """d2 f / d theta2, shape (D, D)  (reference: sensitivity_lib.py:382-383)."""
    _, _, l2 = link(family, linear_predictor(X, theta, offset), y)
    d = scale * w * l2
    return X.T @ (d[:, None] * X) + np.diag(_prior_vec(prior, X.shape[1]))


def cross_hessian(family, X, y, theta, w, prior=None, scale=1.0, offset=None):
    """d/dw (d f / d theta), shape (D, N): column n is scale*ell'_n * x_n
    (reference: sensitivity_lib.py:217-224 wants exactly (len(theta), len(w)))."""
    _, l1, _ = link(family, linear_predictor(X, theta, offset), y)
    return (X * (scale * l1)[:, None]).T


def hvp(family, X, y, theta, w, v, prior=None, scale=1.0, offset=None):
    """Hessian-vector product without forming H (the ``mat_times_vec`` a user of
    solver_lib.get_cg_solver would hand in, solver_lib.py:70-98)."""
    _, _, l2 = link(family, linear_predictor(X, theta, offset), y)
    d = scale * w * l2
    return X.T @ (d * (X @ v)) + _prior_vec(prior, X.shape[1]) * v


def sensitivity(family, X, y, theta, w, prior=None, scale=1.0, offset=None):
    """S = -H^{-1} jac0 with the reference's LAPACK calls (solver_lib.py:27,29;
    sensitivity_lib.py:226).  Returns (H, jac0, S); S has shape (D, N)."""
    H = hessian(family, X, y, theta, w, prior, scale, offset)
    jac0 = cross_hessian(family, X, y, theta, w, prior, scale, offset)
    chol = scipy.linalg.cho_factor(H)
    S = -1 * scipy.linalg.cho_solve(chol, jac0)
    return H, jac0, S


def newton_optimum(family, X, y, w, prior=None, scale=1.0, offset=None, iters=25, tol=1e-13):
    """Plain Newton iteration for theta-hat (the step *before* the hot path: tests
    and the notebook use scipy.optimize, test_sensitivity_lib.py:474-478)."""
    theta = np.zeros(X.shape[1])
    for _ in range(iters):
        g = grad(family, X, y, theta, w, prior, scale, offset)
        H = hessian(family, X, y, theta, w, prior, scale, offset)
        step = scipy.linalg.cho_solve(scipy.linalg.cho_factor(H), g)
        theta = theta - step
        if np.linalg.norm(step) <= tol * max(1.0, np.linalg.norm(theta)):
            break
    return theta


def synth(N, D, family="logistic", seed=0, dtype=np.float64):
    """Synthetic inputs of SURVEY.md section 8(d): X ~ N(0,1), theta* ~ N(0, 1/D),
    y ~ family(x . theta*), w = 1.  (NumPy Generator, PCG64 stream 'seed'.)"""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, D)).astype(dtype)
    theta_star = rng.standard_normal(D) / np.sqrt(D)
    z = X @ theta_star
    if family == "logistic":
        y = (rng.random(N) < _sigmoid(z)).astype(dtype)
    elif family == "poisson":
        y = rng.poisson(np.exp(0.5 * z)).astype(dtype)
    else:
        y = (z + rng.standard_normal(N)).astype(dtype)
    w = np.ones(N, dtype=dtype)
    return X, y, w, theta_star
