"""CPU tests (no GPU): pin the oracle against the reference's own outputs, golden
vectors and known-answer tests.  Tolerances are written next to each check."""
import os
import warnings

import numpy as np
import pytest
import scipy.linalg
import scipy.sparse

from oracle import glm, solvers, sensitivity, block_hessian, lr_cov, vi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


# ---- solver_lib: outputs of the reference itself (tests/golden/make_golden.py) -------------
def test_solvers_match_reference_outputs():
    g = _load("solver_ref.npz")
    h, v, V = g["h"], g["v"], g["V"]
    tol = 1e-14  # same LAPACK / SuperLU calls -> essentially bitwise
    assert np.abs(solvers.get_dense_cholesky_solver(h)(v) - g["dense_v"]).max() <= tol
    assert np.abs(solvers.get_dense_cholesky_solver(h)(V) - g["dense_V"]).max() <= tol
    assert np.abs(solvers.get_cholesky_solver(h)(v) - g["auto_v"]).max() <= tol
    chol = (g["chol_c"], bool(g["chol_lower"]))
    assert np.abs(solvers.get_dense_cholesky_solver(None, chol)(v) - g["prefactored_v"]).max() <= tol
    hs = scipy.sparse.csc_matrix(h)
    assert np.abs(solvers.get_cholesky_solver(hs)(v) - g["sparse_v"]).max() <= tol
    assert np.abs(solvers.get_sparse_cholesky_solver(hs)(V) - g["sparse_V"]).max() <= tol
    with pytest.raises(ValueError, match="must be sparse"):
        solvers.get_sparse_cholesky_solver(h)


def test_solver_kat_seed101():
    """reference tests/test_solver_lib.py:11-43 (decimal=6), including the CG branch the
    reference itself can no longer run on SciPy >= 1.14."""
    np.random.seed(101)
    d = 10
    h = np.random.random((d, d))
    h = h + h.T + d * np.eye(d)
    hs = scipy.sparse.csc_matrix(h)
    v = np.random.random(d)
    ref = np.linalg.solve(h, v)
    for s in (solvers.get_dense_cholesky_solver(h), solvers.get_cholesky_solver(h),
              solvers.get_dense_cholesky_solver(None, scipy.linalg.cho_factor(h)),
              solvers.get_cholesky_solver(hs), solvers.get_sparse_cholesky_solver(hs),
              solvers.get_cg_solver(lambda x: h @ x, d), solvers.get_cg_solver(lambda x: hs @ x, d)):
        np.testing.assert_array_almost_equal(s(v), ref)
    with pytest.warns(UserWarning, match="CG exited with error code 1"):
        solvers.get_cg_solver(lambda x: hs @ x, d, cg_opts={'maxiter': 1})(v)


def test_cg_legacy_matches_modern_scipy():
    """The restated legacy iteration == scipy.sparse.linalg.cg(rtol=tol, atol=0) iterate for iterate."""
    import scipy.sparse.linalg as sla
    rng = np.random.default_rng(5)
    A = rng.standard_normal((60, 40))
    h = A.T @ A + np.eye(40)
    b = rng.standard_normal(40)
    x, info = solvers.cg_legacy(lambda u: h @ u, b, tol=1e-9)
    xs, infos = sla.cg(sla.LinearOperator((40, 40), lambda u: h @ u), b, rtol=1e-9, atol=0.0)
    assert info == infos == 0
    assert np.abs(x - xs).max() <= 1e-13 * np.abs(xs).max()


# ---- GLM closed form vs an independent AD engine and the reference's solver -----------------
def test_glm_closed_form_matches_golden():
    g = _load("glm_small.npz")
    for tag in ("64x5_", "1000x20_"):
        X, y, w, theta = (g[tag + k] for k in ("X", "y", "w", "theta"))
        H, jac0, S = glm.sensitivity("logistic", X, y, theta, w, prior=1.0)
        rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
        assert rel(H, g[tag + "H"]) < 1e-13          # torch.func.hessian
        assert rel(jac0, g[tag + "jac0"]) < 1e-13    # torch.func.jacrev(grad, 1), shape (D, N)
        assert rel(S, g[tag + "S"]) < 1e-13          # reference solver_lib on the AD jac0
        gg = glm.grad("logistic", X, y, theta, w, prior=1.0)
        assert np.abs(gg - g[tag + "g"]).max() < 1e-11
        assert jac0.shape == (X.shape[1], X.shape[0])


@pytest.mark.parametrize("family", glm.FAMILIES)
def test_glm_families_vs_torch_ad(family):
    import torch
    X, y, w, _ = glm.synth(120, 6, family, seed=2)
    rng = np.random.default_rng(0)
    w = 0.5 + rng.random(120)
    offset = 0.1 * rng.standard_normal(120)
    theta = 0.3 * rng.standard_normal(6)
    prior = np.linspace(0.5, 2, 6)
    Xt, yt, ot, pt = (torch.tensor(a) for a in (X, y, offset, prior))

    def f(t, ww):
        z = Xt @ t + ot
        if family == "logistic":
            ell = torch.nn.functional.softplus(z) - yt * z
        elif family == "poisson":
            ell = torch.exp(z) - yt * z
        else:
            ell = 0.5 * (z - yt) ** 2
        return 1.7 * (ww * ell).sum() + 0.5 * (pt * t ** 2).sum()
    tt, wt = torch.tensor(theta), torch.tensor(w)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert abs(glm.value(family, X, y, theta, w, prior, 1.7, offset) - f(tt, wt).item()) < 1e-11 * abs(f(tt, wt).item())
    assert rel(glm.grad(family, X, y, theta, w, prior, 1.7, offset), torch.func.grad(f, 0)(tt, wt).numpy()) < 1e-12
    assert rel(glm.hessian(family, X, y, theta, w, prior, 1.7, offset), torch.func.hessian(f, 0)(tt, wt).numpy()) < 1e-12
    assert rel(glm.cross_hessian(family, X, y, theta, w, prior, 1.7, offset),
               torch.func.jacrev(torch.func.grad(f, 0), 1)(tt, wt).numpy()) < 1e-12
    v = rng.standard_normal(6)
    assert rel(glm.hvp(family, X, y, theta, w, v, prior, 1.7, offset),
               glm.hessian(family, X, y, theta, w, prior, 1.7, offset) @ v) < 1e-13


# ---- linear approximation KATs of the reference tests ---------------------------------------
def test_quadratic_model_kat():
    """tests/test_utils.py:23-75 + tests/test_sensitivity_lib.py:455-554 with
    theta_free = lambda_free = False: H = A, cross-Hessian = I, S = -A^{-1}, prediction exact."""
    dim = 3
    vec = np.linspace(0.1, 0.3, num=dim)
    A = np.outer(vec, vec) + np.eye(dim)
    lam0 = np.linspace(0.5, 10.0, num=dim)
    theta0 = -np.linalg.solve(A, lam0)
    grad = lambda th, lam: A @ th + lam
    for use_h in (False, True):
        for use_c in (False, True):
            s = sensitivity.HyperparameterSensitivityLinearApproximation(
                grad, lambda th, lam: A, lambda th, lam: np.eye(dim), theta0, lam0, validate_optimum=True,
                hessian_at_opt=A if use_h else None, cross_hess_at_opt=np.eye(dim) if use_c else None)
            np.testing.assert_array_almost_equal(s.get_dopt_dhyper(), -np.linalg.inv(A))
            lam1 = lam0 + 0.001
            np.testing.assert_array_almost_equal(
                s.predict_opt_par_from_hyper_par(lam1) - theta0, -np.linalg.solve(A, lam1) - theta0)
    with pytest.raises(ValueError, match="estimating equation is not zero"):
        sensitivity.HyperparameterSensitivityLinearApproximation(
            grad, lambda th, lam: A, lambda th, lam: np.eye(dim), theta0 + 1, lam0, validate_optimum=True)
    with pytest.raises(ValueError, match="hessian_at_opt"):
        sensitivity.HyperparameterSensitivityLinearApproximation(
            grad, None, None, theta0, lam0, hessian_at_opt=np.eye(dim + 1), cross_hess_at_opt=np.eye(dim))
    with pytest.raises(ValueError, match="_estimating_equation_jac0"):
        sensitivity.HyperparameterSensitivityLinearApproximation(
            grad, None, None, theta0, lam0, hessian_at_opt=A, cross_hess_at_opt=np.eye(dim + 1))


def test_weighted_linear_regression_kat():
    """tests/test_sensitivity_lib.py:838-889: f = sum_n w_n (y_n - x_n theta)^2; the first-order
    term equals d/dw of the closed form theta(w) = (X^T W X)^{-1} X^T W y."""
    rng = np.random.default_rng(0)
    n_obs, dim = 10, 2
    x = rng.random((n_obs, dim))
    y = x @ np.array([0.5, -0.1]) + rng.normal(size=n_obs)
    run = lambda w: np.linalg.solve(np.einsum('n,ni,nj->ij', w, x, x), np.einsum('n,ni,n->i', w, x, y))
    w1 = np.ones(n_obs)
    theta0 = run(w1)
    H, jac0, S = glm.sensitivity("gaussian", x, y, theta0, w1, prior=None, scale=2.0)
    assert np.linalg.norm(glm.grad("gaussian", x, y, theta0, w1, None, 2.0)) < 1e-8
    dw = rng.random(n_obs) - 0.5
    eps = 1e-6
    d1_fd = (run(w1 + eps * dw) - run(w1 - eps * dw)) / (2 * eps)
    np.testing.assert_array_almost_equal(S @ dw, d1_fd)
    resid = y - x @ theta0
    xtx = np.einsum('n,ni,nj->ij', w1, x, x)
    S_closed = np.linalg.solve(xtx, (x * resid[:, None]).T)      # SURVEY.md section 4 closed form
    assert np.abs(S - S_closed).max() < 1e-12


def test_notebook_outputs_kat():
    """The reference's own printed outputs (notebook cells 16 and 24): optimum and the
    approximate leave-one-out (n = 10) parameters, 8 printed decimals."""
    g = _load("notebook_kat.npz")
    x, free0, H, J = g["x"], g["free0"], g["H"], g["J"]
    N = x.shape[0]
    s = sensitivity.HyperparameterSensitivityLinearApproximation(
        None, None, None, free0, np.ones(N), hessian_at_opt=H, cross_hess_at_opt=J)
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
    s0, m0 = fold(free0)
    s1, m1 = fold(free1)
    tol = 5e-8   # 8 printed decimals; the notebook's optimum came from trust-ncg with gtol 1e-12
    assert np.abs(s0 - g["printed_opt_sigma"]).max() < tol and np.abs(m0 - g["printed_opt_mu"]).max() < tol
    assert np.abs(s1 - g["printed_loo10_sigma"]).max() < tol and np.abs(m1 - g["printed_loo10_mu"]).max() < tol


# ---- LR covariance and block Hessian ----------------------------------------------------------
def test_lr_covariance_kat():
    """tests/test_lr_cov_lib.py:16-135 with free=False: H = blockdiag(Sigma^{-1}, diag(.5/var^2));
    LR covariance of the means == Sigma exactly."""
    dim = 4
    mu = np.arange(0, dim).astype(float)
    cov = dim * np.eye(dim) + np.outer(mu, mu)
    info = np.linalg.inv(cov)
    var = 1 / np.diag(info)
    opt = np.concatenate([mu, var])
    H = scipy.linalg.block_diag(info, np.diag(0.5 / var ** 2))
    grad = lambda eta: np.concatenate([info @ (eta[:dim] - mu), 0.5 * np.diag(info) - 0.5 / eta[dim:]])
    for init_h in (False, True):
        lr = lr_cov.LinearResponseCovariances(grad, lambda eta: H, opt, validate_optimum=True,
                                              hessian_at_opt=H if init_h else None, grad_tol=1e-15)
        np.testing.assert_array_almost_equal(H, lr.get_hessian_at_opt())
        J = np.hstack([np.eye(dim), np.zeros((dim, dim))])
        c = lr.get_lr_covariance_from_jacobians(J, J)
        np.testing.assert_array_almost_equal(cov, c)
        np.testing.assert_array_almost_equal(c[0:2, 2:4], lr.get_lr_covariance_from_jacobians(J[0:2], J[2:4]))
        for bad in ((J.T, J), (J, J.T), (J[:, :, None], J), (J, J[:, :, None])):
            with pytest.raises(ValueError):
                lr.get_lr_covariance_from_jacobians(*bad)
    with pytest.raises(ValueError, match="gradient is not zero"):
        lr_cov.LinearResponseCovariances(grad, lambda eta: H, opt + 0.1, validate_optimum=True)


def test_block_hessian_structure():
    """tests/test_sparse_hessian_lib.py:50-53, 101-113: block == dense for a separable objective;
    block + global == dense with global parameters; COO order and duplicate rules as the reference."""
    G, ng, k = 6, 5, 2
    X, y = vi.synth(G, ng, k, seed=3)
    m = vi.HierGaussMFVI(X, y, G, ng, k, tau_y=2.0, tau_th=1.5, tau_mu=0.5)
    eta = m.optimum() * 1.05
    Hd = m.hessian(eta)
    sb = block_hessian.SparseBlockHessian(lambda o, v: m.hvp(o, v), m.sparsity_array())
    blk = sb.get_block_hessian(eta)
    assert blk.nnz == G * (2 * k) ** 2
    M = 2 * k
    # reference order: ib outer, block middle, row-in-block inner (sparse_hessian_lib.py:93-102)
    assert list(blk.col[:M]) == [0] * M and list(blk.row[:M]) == list(range(M))
    full = (blk + sb.get_global_hessian(eta)).todense()
    np.testing.assert_array_almost_equal(np.asarray(full), Hd)
    full2 = (blk + sb.get_global_hessian(eta, global_inds=m.global_inds())).todense()
    np.testing.assert_array_almost_equal(np.asarray(full2), Hd)
    np.testing.assert_array_almost_equal(np.asarray(sb.get_hessian(eta).todense()), Hd)
    with pytest.raises(ValueError, match="must be unique"):
        block_hessian.SparseBlockHessian(lambda o, v: v, np.array([[0, 1], [1, 2]]))
    with pytest.raises(ValueError, match="must be disjoint"):
        sb.get_global_hessian(eta, global_inds=np.array([0]))
    with pytest.raises(ValueError, match="must be a vector"):
        sb.get_block_hessian(eta[None, :])


def test_vi_model_exactness():
    """LR covariance of the variational means == exact posterior covariance (Gaussian model):
    the property tests/test_lr_cov_lib.py:90-93 relies on."""
    G, ng, k = 4, 6, 2
    X, y = vi.synth(G, ng, k, seed=1)
    m = vi.HierGaussMFVI(X, y, G, ng, k, tau_y=1.3, tau_th=0.7, tau_mu=0.4)
    eta = m.optimum()
    assert np.linalg.norm(m.grad(eta)) < 1e-10
    H = m.hessian(eta)
    # exact joint posterior precision of (theta_1..theta_G, mu)
    P = np.zeros((k * (G + 1), k * (G + 1)))
    Xg = X.reshape(G, ng, k)
    for g in range(G):
        P[g * k:(g + 1) * k, g * k:(g + 1) * k] = 1.3 * Xg[g].T @ Xg[g] + 0.7 * np.eye(k)
        P[g * k:(g + 1) * k, G * k:] = -0.7 * np.eye(k)
        P[G * k:, g * k:(g + 1) * k] = -0.7 * np.eye(k)
    P[G * k:, G * k:] = (G * 0.7 + 0.4) * np.eye(k)
    mean_idx = np.concatenate([np.arange(g * 2 * k, g * 2 * k + k) for g in range(G)] +
                              [np.arange(2 * k * G, 2 * k * G + k)])
    J = np.zeros((len(mean_idx), m.d))
    J[np.arange(len(mean_idx)), mean_idx] = 1.0
    lr = lr_cov.LinearResponseCovariances(m.grad, m.hessian, eta, validate_optimum=True, grad_tol=1e-8)
    cov = lr.get_lr_covariance_from_jacobians(J, J)
    assert np.abs(cov - np.linalg.inv(P)).max() < 1e-12


# ---- second-order data-weight terms -----------------------------
def test_taylor_oracle_weighted_regression_kat():
    """tests/test_sensitivity_lib.py:838-901: Taylor terms of theta(w) for weighted least squares
    equal the directional derivatives of the closed form (X^T W X)^{-1} X^T W y -- here through
    torch.func.jvp (nested once) in the role of the reference's `_append_jvp`; 6 decimals there,
    1e-10 here."""
    import torch
    from oracle import taylor
    rng = np.random.default_rng(0)
    n_obs, dim = 10, 2
    x = rng.random((n_obs, dim))
    y = x @ np.array([0.5, -0.1]) + rng.normal(size=n_obs)
    xt, yt = torch.tensor(x), torch.tensor(y)

    def run(w):
        # (X^T W X)^{-1} X^T W y by Cramer's rule (dim = 2): nested forward-mode AD through
        # torch.linalg.solve returns HALF the second derivative (checked against finite differences)
        a = torch.einsum('n,ni,nj->ij', w, xt, xt)
        b = torch.einsum('n,ni,n->i', w, xt, yt)
        det = a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]
        return torch.stack([a[1, 1] * b[0] - a[0, 1] * b[1], a[0, 0] * b[1] - a[1, 0] * b[0]]) / det
    w1 = torch.ones(n_obs, dtype=torch.float64)
    dw = torch.tensor(rng.random(n_obs) - 0.5)
    theta0 = run(w1).numpy()
    d1_fn = lambda w: torch.func.jvp(run, (w,), (dw,))[1]
    d1 = d1_fn(w1).numpy()
    d2 = torch.func.jvp(d1_fn, (w1,), (dw,))[1].numpy()
    d2_fn = lambda w: torch.func.jvp(d1_fn, (w,), (dw,))[1]
    d3_fn = lambda w: torch.func.jvp(d2_fn, (w,), (dw,))[1]
    d3 = d3_fn(w1).numpy()
    d4 = torch.func.jvp(d3_fn, (w1,), (dw,))[1].numpy()
    got = taylor.input_derivs("gaussian", x, y, theta0, w1.numpy(), dw.numpy(), prior=None, scale=2.0, max_order=4)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(got[0], d1) < 1e-10 and rel(got[1], d2) < 1e-10
    assert rel(got[2], d3) < 1e-10 and rel(got[3], d4) < 1e-10          # orders 3 and 4 of the reference test
    series = taylor.taylor_series("gaussian", x, y, theta0, w1.numpy(), (w1 + dw).numpy(), None, 2.0)
    assert rel(series, theta0 + d1 + d2 / 2.0) < 1e-12


@pytest.mark.parametrize("family", glm.FAMILIES)
def test_taylor_oracle_vs_torch_ad_of_the_estimating_equation(family):
    """The closed-form order-2 right-hand side == nested forward-mode AD of grad_theta f (what the
    reference's DerivativeTerm machinery evaluates), and the resulting d2 == a central second
    difference of the re-optimised optimum along w + t dw (loose: 1e-5)."""
    import torch
    from oracle import taylor
    N, D = 150, 5
    X, y, w, _ = glm.synth(N, D, family, seed=4)
    rng = np.random.default_rng(1)
    w = 0.5 + rng.random(N)
    dw = 0.2 * (rng.random(N) - 0.5)
    prior = np.linspace(0.5, 2, D)
    theta = glm.newton_optimum(family, X, y, w, prior=prior, scale=1.3)
    Xt, yt, pt = (torch.tensor(a) for a in (X, y, prior))

    def f(t, ww):
        z = Xt @ t
        if family == "logistic":
            ell = torch.nn.functional.softplus(z) - yt * z
        elif family == "poisson":
            ell = torch.exp(z) - yt * z
        else:
            ell = 0.5 * (z - yt) ** 2
        return 1.3 * (ww * ell).sum() + 0.5 * (pt * t ** 2).sum()
    G = torch.func.grad(f, 0)
    d1, d2, d3 = taylor.input_derivs(family, X, y, theta, w, dw, prior, 1.3, max_order=3)
    tt, wt, dwt, d1t = (torch.tensor(a) for a in (theta, w, dw, d1))
    # second total derivative of G along (theta + s d1, w + s dw) minus the H d2 term
    path = lambda s: G(tt + s * d1t, wt + s * dwt)
    first = lambda s: torch.func.jvp(path, (s,), (torch.ones((), dtype=torch.float64),))[1]
    second = torch.func.jvp(first, (torch.zeros((), dtype=torch.float64),), (torch.ones((), dtype=torch.float64),))[1]
    H = glm.hessian(family, X, y, theta, w, prior, 1.3)
    rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
    assert rel(-H @ d2, second.numpy()) < 1e-10
    eps = 1e-3
    opt = lambda ww: glm.newton_optimum(family, X, y, ww, prior=prior, scale=1.3, iters=50)
    fd2 = (opt(w + eps * dw) - 2 * theta + opt(w - eps * dw)) / eps ** 2
    assert rel(d2, fd2) < 1e-5
    # order 3: third total derivative of G along theta + s d1 + s^2/2 d2, w + s dw (nested forward-mode AD)
    # equals -H d3; and d3 == a central third difference of the re-optimised optimum (loose)
    d2t = torch.tensor(d2)
    one = torch.ones((), dtype=torch.float64)
    path3 = lambda s_: G(tt + s_ * d1t + 0.5 * s_ * s_ * d2t, wt + s_ * dwt)
    f1 = lambda s_: torch.func.jvp(path3, (s_,), (one,))[1]
    f2 = lambda s_: torch.func.jvp(f1, (s_,), (one,))[1]
    third = torch.func.jvp(f2, (torch.zeros((), dtype=torch.float64),), (one,))[1]
    assert rel(-H @ d3, third.numpy()) < 1e-10
    if family != "gaussian":
        e3 = 5e-2
        fd3 = (opt(w + 2 * e3 * dw) - 2 * opt(w + e3 * dw) + 2 * opt(w - e3 * dw) - opt(w - 2 * e3 * dw)) / (2 * e3 ** 3)
        assert rel(d3, fd3) < 5e-3


@pytest.mark.parametrize("family", ["logistic", "poisson"])
def test_taylor_oracle_any_order_against_refits(family):
    """oracle.taylor.input_derivs_general (power-series composition) is pinned two ways: it reproduces the closed
    forms of orders 1-3 (also under max_input_order), and the order-K series misses a RE-FITTED optimum by
    O(eps^(K+1)) -- halving eps divides the error by 2^(K+1) for K = 3..6."""
    from math import factorial
    from oracle import glm, taylor
    N, D = 400, 5
    X, y, w, _ = glm.synth(N, D, family, seed=3)
    if family == "poisson":
        X = 0.3 * X
    rng = np.random.default_rng(0)
    w = rng.uniform(0.5, 1.5, N)
    dw = 0.5 * rng.standard_normal(N)
    theta = glm.newton_optimum(family, X, y, w, prior=1.0)
    gen = taylor.input_derivs_general(family, X, y, theta, w, dw, prior=1.0, max_order=6)
    for mio in (None, 1, 2):
        closed = taylor.input_derivs(family, X, y, theta, w, dw, prior=1.0, max_order=3, max_input_order=mio)
        g = gen if mio is None else taylor.input_derivs_general(family, X, y, theta, w, dw, prior=1.0, max_order=3,
                                                                 max_input_order=mio)
        for k in range(3):
            assert np.abs(closed[k] - g[k]).max() <= 1e-13 * np.abs(closed[k]).max()
    for K in (3, 4, 5, 6):
        errs = []
        for eps in (0.2, 0.1):
            refit = glm.newton_optimum(family, X, y, w + eps * dw, prior=1.0)
            series = theta + sum(gen[j] * eps ** (j + 1) / factorial(j + 1) for j in range(K))
            errs.append(np.abs(series - refit).max())
        assert 0.8 * 2 ** (K + 1) < errs[0] / errs[1] < 1.25 * 2 ** (K + 1), (K, errs)
