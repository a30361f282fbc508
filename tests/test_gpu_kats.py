"""GPU parity against the committed golden fixtures and the reference's known-answer tests,
through the drop-in classes (HIP path).  Tolerances written next to each check."""
import os

import numpy as np
import pytest
import scipy.linalg
import scipy.sparse

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_solver_lib_matches_reference_outputs(relerr):
    """solver_ref.npz holds the outputs of /root/reference/vittles/solver_lib.py itself."""
    import vittles_amd as va
    g = _load("solver_ref.npz")
    h, v, V = g["h"], g["v"], g["V"]
    tol = 1e-12
    sl = va.solver_lib
    assert relerr(sl.get_dense_cholesky_solver(h)(v), g["dense_v"]) < tol
    out = sl.get_dense_cholesky_solver(h)(V)
    assert out.shape == V.shape and relerr(out, g["dense_V"]) < tol
    assert relerr(sl.get_cholesky_solver(h)(v), g["auto_v"]) < tol
    chol = (g["chol_c"], bool(g["chol_lower"]))
    assert relerr(sl.get_dense_cholesky_solver(None, chol)(v), g["prefactored_v"]) < tol
    hs = scipy.sparse.csc_matrix(h)
    assert relerr(sl.get_cholesky_solver(hs)(v), g["sparse_v"]) < tol
    assert relerr(sl.get_sparse_cholesky_solver(hs)(V), g["sparse_V"]) < tol
    # lower-triangular scipy factor is accepted too
    cl = scipy.linalg.cho_factor(h, lower=True)
    assert relerr(sl.get_dense_cholesky_solver(None, cl)(v), g["dense_v"]) < tol


def test_glm_golden(relerr):
    import vittles_amd as va
    g = _load("glm_small.npz")
    for tag in ("64x5_", "1000x20_"):
        X, y, w, theta = (g[tag + k] for k in ("X", "y", "w", "theta"))
        obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
        s = va.HyperparameterSensitivityLinearApproximation(obj, theta, w, validate_optimum=True, grad_tol=1e-8)
        assert relerr(s.get_hessian_at_opt(), g[tag + "H"]) < 1e-10
        assert relerr(s.get_dopt_dhyper(), g[tag + "S"]) < 1e-10
        # explicit Hessian / cross-Hessian arguments take the generic dense path
        s2 = va.HyperparameterSensitivityLinearApproximation(
            obj, theta, w, hessian_at_opt=g[tag + "H"], cross_hess_at_opt=g[tag + "jac0"])
        assert relerr(s2.get_dopt_dhyper(), g[tag + "S"]) < 1e-10


def test_quadratic_model_kat():
    """tests/test_utils.py:23-75 / test_sensitivity_lib.py:455-554 (non-free variant), 6 decimals
    as the reference; the objective is a plain callable with both derivative matrices supplied."""
    import vittles_amd as va
    dim = 3
    vec = np.linspace(0.1, 0.3, num=dim)
    A = np.outer(vec, vec) + np.eye(dim)
    lam0 = np.linspace(0.5, 10.0, num=dim)
    theta0 = -np.linalg.solve(A, lam0)
    f = lambda th, lam: 0.5 * th @ A @ th + lam @ th
    s = va.HyperparameterSensitivityLinearApproximation(
        f, theta0, lam0, hessian_at_opt=A, cross_hess_at_opt=np.eye(dim))
    np.testing.assert_array_almost_equal(s.get_dopt_dhyper(), -np.linalg.inv(A))
    lam1 = lam0 + 0.001
    np.testing.assert_array_almost_equal(s.predict_opt_par_from_hyper_par(lam1), -np.linalg.solve(A, lam1))
    assert s.get_hessian_at_opt() is A                       # returned by reference (:412)
    with pytest.raises(ValueError, match="hessian_at_opt"):
        va.HyperparameterSensitivityLinearApproximation(
            f, theta0, lam0, hessian_at_opt=np.eye(dim + 1), cross_hess_at_opt=np.eye(dim))
    with pytest.raises(ValueError, match="_estimating_equation_jac0"):
        va.HyperparameterSensitivityLinearApproximation(
            f, theta0, lam0, hessian_at_opt=A, cross_hess_at_opt=np.eye(dim + 1))


def test_weighted_linear_regression_kat(relerr):
    """tests/test_sensitivity_lib.py:838-889 as a registered gaussian objective with scale 2."""
    import vittles_amd as va
    rng = np.random.default_rng(0)
    n_obs, dim = 10, 2
    x = rng.random((n_obs, dim))
    y = x @ np.array([0.5, -0.1]) + rng.normal(size=n_obs)
    w1 = np.ones(n_obs)
    theta0 = np.linalg.solve(x.T @ x, x.T @ y)
    obj = va.WeightedGLMObjective(x, y, family="gaussian", scale=2.0)
    s = va.HyperparameterSensitivityLinearApproximation(obj, theta0, w1, validate_optimum=True)
    resid = y - x @ theta0
    S_closed = np.linalg.solve(x.T @ x, (x * resid[:, None]).T)
    assert relerr(s.get_dopt_dhyper(), S_closed) < 1e-10
    assert abs(obj(theta0, w1) - np.sum(resid ** 2)) < 1e-12


def test_notebook_outputs_kat():
    """Numbers printed by the reference's notebook (cells 16, 24), reproduced on the GPU."""
    import vittles_amd as va
    g = _load("notebook_kat.npz")
    free0, H, J = g["free0"], g["H"], g["J"]
    N = J.shape[1]
    s = va.HyperparameterSensitivityLinearApproximation(
        lambda t, w: 0.0, free0, np.ones(N), hessian_at_opt=H, cross_hess_at_opt=J)
    w = np.ones(N)
    w[10] = 0
    free1 = s.predict_opt_par_from_hyper_par(w)

    def fold(free):
        L = np.zeros((3, 3))
        idx = 0
        for i in range(3):
            for j in range(i + 1):
                L[i, j] = np.exp(free[idx]) if i == j else free[idx]
                idx += 1
        return L @ L.T, free[6:9]
    s1, m1 = fold(free1)
    assert np.abs(s1 - g["printed_loo10_sigma"]).max() < 5e-8
    assert np.abs(m1 - g["printed_loo10_mu"]).max() < 5e-8
