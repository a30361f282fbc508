"""GPU parity of the SURVEY.md section 8(f) rows built after the core path:
  rank 2 -- the optimiser in front of the path (`vittles_amd.optim.newton_glm`) against the oracle's
            Newton iteration and SciPy's minimiser (what the reference's tests use,
            vittles/tests/test_sensitivity_lib.py:474-478);
  rank 3 -- the differentiable optimum `get_opt_par_function` (vittles/sensitivity_lib.py:18-82,
            249-254, 426-429) against the reference's own test of it
            (vittles/tests/test_sensitivity_lib.py:556-598), with torch autograd in the role of autograd.
Tolerances are written next to each check."""
import numpy as np
import pytest
import scipy.optimize

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- rank 2: optimiser -------
@pytest.mark.parametrize("family", ["logistic", "poisson", "gaussian"])
@pytest.mark.parametrize("N,D", [(500, 5), (5000, 20), (20000, 100)])
def test_newton_glm_matches_oracle_optimum(family, N, D, relerr):
    import vittles_amd as va
    from oracle import glm
    X, y, w, _ = glm.synth(N, D, family, seed=N + D)
    rng = np.random.default_rng(1)
    w = 0.5 + rng.random(N)
    prior = np.linspace(0.5, 2.0, D)
    ref = glm.newton_optimum(family, X, y, w, prior=prior)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior)
    theta, info = va.optim.newton_glm(obj, w)
    assert isinstance(theta, np.ndarray) and info["converged"]
    assert info["grad_norm"] <= 1e-9
    assert relerr(theta, ref) < 1e-10
    # the optimum passes the reference's own validation (sensitivity_lib.py:203-215)
    va.HyperparameterSensitivityLinearApproximation(obj, theta, w, validate_optimum=True, grad_tol=1e-8)


def test_newton_glm_vs_scipy_minimize_and_line_search(relerr):
    """SciPy's optimiser on the oracle objective (the reference's way of getting theta-hat);
    a far-away start makes the pure Newton step overshoot, so the backtracking runs."""
    import torch
    import vittles_amd as va
    from oracle import glm
    N, D = 3000, 6
    X, y, w, _ = glm.synth(N, D, "logistic", seed=3)
    f = lambda t: glm.value("logistic", X, y, t, w, prior=1e-3)
    g = lambda t: glm.grad("logistic", X, y, t, w, prior=1e-3)
    res = scipy.optimize.minimize(f, np.zeros(D), jac=g, method="BFGS", options=dict(gtol=1e-10))
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1e-3)
    # saturated start: the full Newton step (length 23) raises the objective eight-fold
    start = torch.full((D,), 2.0, dtype=torch.float64, device="cuda")
    theta, info = va.optim.newton_glm(obj, torch.tensor(w, device="cuda"), opt_par_init=start)
    assert isinstance(theta, torch.Tensor) and theta.is_cuda and info["converged"]
    assert info["halvings"] > 0
    assert relerr(theta.cpu().numpy(), res.x) < 1e-6          # BFGS's own accuracy
    assert abs(info["value"] - res.fun) <= 1e-9 * abs(res.fun)


# ---------------------------------------------------------------- rank 3: differentiable optimum -------
def _quadratic_la(va, dim=3):
    vec = np.linspace(0.1, 0.3, num=dim)
    A = np.outer(vec, vec) + np.eye(dim)
    lam0 = np.linspace(0.5, 10.0, num=dim)
    theta0 = -np.linalg.solve(A, lam0)
    f = lambda th, lam: 0.5 * th @ A @ th + lam @ th
    la = va.HyperparameterSensitivityLinearApproximation(f, theta0, lam0, hessian_at_opt=A,
                                                         cross_hess_at_opt=np.eye(dim))
    return la, theta0, lam0


def test_opt_par_function_reference_test_quadratic_model():
    """test_sensitivity_lib.py:556-598 on the QuadraticModel (non-free), 6 decimals as there."""
    import torch
    import torch.autograd.forward_ad as fwad
    import vittles_amd as va
    la, theta0, lam0 = _quadratic_la(va)
    dim = len(lam0)
    get_opt_par = la.get_opt_par_function()
    np.testing.assert_array_almost_equal(theta0, get_opt_par(lam0))
    with pytest.raises(ValueError):
        get_opt_par(lam0 + 1)
    lam_t = torch.tensor(lam0, device="cuda", requires_grad=True)
    with pytest.raises(ValueError):
        get_opt_par(lam_t + 1)

    def fun_of_opt(lam):
        return torch.exp(torch.sum(get_opt_par(lam) + 0.1))

    dopt_dhyper = la.get_dopt_dhyper()
    val = float(fun_of_opt(lam_t))
    # reverse mode
    (grad,) = torch.autograd.grad(fun_of_opt(lam_t), lam_t)
    np.testing.assert_array_almost_equal(dopt_dhyper.T @ np.full(dim, val), grad.cpu().numpy())
    # forward mode
    delta = np.random.default_rng(0).random(dim)
    with fwad.dual_level():
        dual = fwad.make_dual(lam_t.detach(), torch.tensor(delta, device="cuda"))
        tangent = fwad.unpack_dual(fun_of_opt(dual)).tangent
    np.testing.assert_array_almost_equal(delta @ dopt_dhyper.T @ np.full(dim, val), float(tangent))
    # higher orders fail (the reference raises NotImplementedError; torch's once_differentiable RuntimeError)
    out = fun_of_opt(lam_t)
    (g1,) = torch.autograd.grad(out, lam_t, create_graph=True)
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.autograd.grad(g1.sum(), lam_t)


def test_opt_par_function_glm_vjp_jvp(relerr):
    """Data weights as the hyper-parameter: VJP = g^T S (N,), JVP = S t (D,) at 1e-10 against the
    oracle's dense S."""
    import torch
    import torch.autograd.forward_ad as fwad
    import vittles_amd as va
    from oracle import glm
    N, D = 6000, 40
    X, y, w, _ = glm.synth(N, D, "logistic", seed=9)
    theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)
    _, _, S_ref = glm.sensitivity("logistic", X, y, theta, w, prior=1.0)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    w_t = torch.tensor(w, device="cuda", requires_grad=True)
    la = va.HyperparameterSensitivityLinearApproximation(obj, torch.tensor(theta, device="cuda"), w_t.detach())
    fn = la.get_opt_par_function()
    out = fn(w_t)
    assert relerr(out.detach().cpu().numpy(), theta) == 0.0
    c = np.random.default_rng(2).standard_normal(D)
    (gw,) = torch.autograd.grad((out * torch.tensor(c, device="cuda")).sum(), w_t)
    assert relerr(gw.cpu().numpy(), c @ S_ref) < 1e-10
    t = np.random.default_rng(3).standard_normal(N)
    with fwad.dual_level():
        dual = fwad.make_dual(w_t.detach(), torch.tensor(t, device="cuda"))
        tangent = fwad.unpack_dual(fn(dual)).tangent
    assert relerr(tangent.cpu().numpy(), S_ref @ t) < 1e-10


# ---------------------------------------------------------------- rank 4: second-order data-weight terms -------
@pytest.mark.parametrize("family", ["logistic", "poisson", "gaussian"])
@pytest.mark.parametrize("N,D", [(300, 4), (5000, 33), (20000, 100)])
def test_taylor_expansion_orders_1_2_match_oracle(family, N, D, relerr):
    """ParametricSensitivityTaylorExpansion.optimization_objective for data weights
    (vittles/sensitivity_lib.py:1021-1363) against oracle/taylor.py, 1e-10."""
    import vittles_amd as va
    from oracle import glm, taylor
    X, y, w, _ = glm.synth(N, D, family, seed=N + 1)
    rng = np.random.default_rng(5)
    w = 0.5 + rng.random(N)
    dw = rng.random(N) - 0.5
    prior = np.linspace(0.5, 2.0, D)
    theta = glm.newton_optimum(family, X, y, w, prior=prior, scale=1.5)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior, scale=1.5)
    te = va.ParametricSensitivityTaylorExpansion.optimization_objective(
        objective_function=obj, input_val0=theta, hyper_val0=w, order=3)
    assert te.get_max_order() == 3
    d_ref = taylor.input_derivs(family, X, y, theta, w, dw, prior, 1.5, max_order=3)
    d_got = te.evaluate_input_derivs(dw)
    assert len(d_got) == 3 and all(isinstance(d, np.ndarray) for d in d_got)
    if np.abs(d_ref[2]).max() > 0:
        assert relerr(d_got[2], d_ref[2]) < 1e-10
    assert len(te.evaluate_input_derivs(dw, max_order=2)) == 2
    assert relerr(d_got[0], d_ref[0]) < 1e-10
    if np.abs(d_ref[1]).max() > 0:
        assert relerr(d_got[1], d_ref[1]) < 1e-10
    for mo in (1, 2, 3):
        ref = taylor.taylor_series(family, X, y, theta, w, w + dw, prior, 1.5, max_order=mo)
        assert relerr(te.evaluate_taylor_series(w + dw, max_order=mo), ref) < 1e-10
    assert relerr(te.evaluate_taylor_series(w + dw, add_offset=False),
                  taylor.taylor_series(family, X, y, theta, w, w + dw, prior, 1.5, add_offset=False, max_order=3)) < 1e-10
    # the first-order term is the linear approximation's prediction (sensitivity_lib.py:245-247)
    la = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
    assert relerr(te.evaluate_taylor_series(w + dw, max_order=1), la.predict_opt_par_from_hyper_par(w + dw)) < 1e-10
    with pytest.raises(ValueError):
        te.evaluate_taylor_series(w + dw, max_order=4)
    with pytest.raises(ValueError):
        te.evaluate_taylor_series(w + dw, max_order=0)


@pytest.mark.parametrize("family", ["logistic", "poisson"])
def test_taylor_expansion_max_input_order_truncates_like_the_reference(family, relerr, capsys):
    """the reference drops every term with more than ``max_input_order`` partial derivatives w.r.t.
    the input (vittles/sensitivity_lib.py:1247-1249); orders 2 and 3 then differ from the untruncated series."""
    import vittles_amd as va
    from oracle import glm, taylor
    N, D = 3000, 17
    X, y, w, _ = glm.synth(N, D, family, seed=77)
    rng = np.random.default_rng(6)
    w = 0.5 + rng.random(N)
    dw = rng.random(N) - 0.5
    theta = glm.newton_optimum(family, X, y, w, prior=1.0)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=1.0)
    full = taylor.input_derivs(family, X, y, theta, w, dw, 1.0, max_order=3)
    for mio in (1, 2, 3):
        te = va.ParametricSensitivityTaylorExpansion.optimization_objective(obj, theta, w, order=3,
                                                                            max_input_order=mio)
        ref = taylor.input_derivs(family, X, y, theta, w, dw, 1.0, max_order=3, max_input_order=mio)
        got = te.evaluate_input_derivs(dw)
        for k in range(3):
            assert relerr(got[k], ref[k]) < 1e-10, (mio, k)
        if mio < 3:          # the truncation really changes the result
            assert relerr(ref[2], full[2]) > 1e-6
    te.print_terms()
    out = capsys.readouterr().out
    assert "Terms for order 3:" in out and "eps[1]" in out
    with pytest.raises(ValueError):
        te.print_terms(k=4)


@pytest.mark.parametrize("family,N,D", [("logistic", 3000, 17), ("poisson", 2000, 9), ("logistic", 20000, 100),
                                        ("poisson", 700, 300)])
def test_taylor_expansion_higher_orders_match_oracle(family, N, D, relerr, capsys):
    """Orders 4..8 for the non-Gaussian families (the reference expands to any order, and its own max-order test
    runs at order 4: vittles/tests/test_sensitivity_lib.py:770-773): vt_glm_taylor_coeff (partial Bell polynomials by
    recurrence, logistic derivatives by P_{m+1} = P_m' (s - s^2)) against the oracle's power-series composition with
    the Stirling-number closed form of the logistic derivatives; and the order-K series against a re-fit."""
    import vittles_amd as va
    from oracle import glm, taylor
    X, y, w, _ = glm.synth(N, D, family, seed=N + D)
    if family == "poisson":
        X = 0.3 * X
    rng = np.random.default_rng(8)
    w = 0.5 + rng.random(N)
    dw = rng.random(N) - 0.5
    theta = glm.newton_optimum(family, X, y, w, prior=1.0, scale=1.2)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=1.0, scale=1.2)
    K = 8
    te = va.ParametricSensitivityTaylorExpansion.optimization_objective(obj, theta, w, order=K)
    got = te.evaluate_input_derivs(dw)
    ref = taylor.input_derivs(family, X, y, theta, w, dw, 1.0, 1.2, max_order=K)
    assert len(got) == K
    for k in range(K):
        assert relerr(got[k], ref[k]) < 1e-9, (k + 1, relerr(got[k], ref[k]))
    # a second call (other direction) must not reuse the cached zeta_j of the first
    dw2 = rng.random(N) - 0.5
    got2 = te.evaluate_input_derivs(dw2, max_order=5)
    ref2 = taylor.input_derivs(family, X, y, theta, w, dw2, 1.0, 1.2, max_order=5)
    assert relerr(got2[4], ref2[4]) < 1e-9
    # the series converges on the re-fitted optimum: order 6 beats order 3 by at least two orders of magnitude
    eps = 0.25
    refit = glm.newton_optimum(family, X, y, w + eps * dw, prior=1.0, scale=1.2)
    e3 = relerr(te.evaluate_taylor_series(w + eps * dw, max_order=3), refit)
    e6 = relerr(te.evaluate_taylor_series(w + eps * dw, max_order=6), refit)
    assert e6 < 1e-2 * e3 or e6 < 1e-13, (e3, e6)
    # truncation in the input order, as the reference's order-4 test uses it (max_input_order 1 and 2)
    for mio in (1, 2, 3):
        tem = va.ParametricSensitivityTaylorExpansion.optimization_objective(obj, theta, w, order=5, max_input_order=mio)
        refm = taylor.input_derivs(family, X, y, theta, w, dw, 1.0, 1.2, max_order=5, max_input_order=mio)
        gotm = tem.evaluate_input_derivs(dw)
        for k in range(5):
            assert relerr(gotm[k], refm[k]) < 1e-9, (mio, k + 1)
        assert relerr(refm[3], ref[3]) > 1e-6
    tem.print_terms(k=4)
    out = capsys.readouterr().out
    assert "Terms for order 4:" in out and "12 * eta[1, 1, 0, 0] * eps[1]" in out and "eta[4, 0, 0, 0]" not in out
    with pytest.raises(NotImplementedError):
        va.ParametricSensitivityTaylorExpansion.optimization_objective(obj, theta, w, order=9)


def test_taylor_expansion_reference_regression_kat():
    """vittles/tests/test_sensitivity_lib.py:838-901 (orders 1 and 2 of its four), 6 decimals as there:
    f = sum_n w_n (y_n - x_n theta)^2 is the gaussian family with scale 2 and no prior."""
    import torch
    import vittles_amd as va
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
    obj = va.WeightedGLMObjective(x, y, family="gaussian", prior=None, scale=2.0)
    te = va.ParametricSensitivityTaylorExpansion.optimization_objective(
        objective_function=obj, input_val0=theta0, hyper_val0=w1.numpy(), order=4,
        max_hyper_order=1, max_input_order=2, forward_mode=False)
    w_new = (w1 + dw).numpy()
    np.testing.assert_array_almost_equal(te.evaluate_taylor_series(w_new, max_order=1), theta0 + d1)
    np.testing.assert_array_almost_equal(te.evaluate_taylor_series(w_new, max_order=2), theta0 + d1 + d2 / 2.0)
    np.testing.assert_array_almost_equal(te.evaluate_taylor_series(w_new, max_order=3),
                                         theta0 + d1 + d2 / 2.0 + d3 / 6.0)
    np.testing.assert_array_almost_equal(te.evaluate_taylor_series(w_new, max_order=4),
                                         theta0 + d1 + d2 / 2.0 + d3 / 6.0 + d4 / 24.0)
    lg = va.WeightedGLMObjective(x, (y > 0).astype(float), family="logistic", prior=1.0)
    with pytest.raises(NotImplementedError):          # non-gaussian families stop at order 8 (round 3; was 3)
        va.ParametricSensitivityTaylorExpansion.optimization_objective(lg, theta0, w1.numpy(), order=9)
