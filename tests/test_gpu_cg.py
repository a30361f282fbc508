"""GPU parity of the matrix-free CG path (K6) against the oracle's restatement of the same
iteration and against the reference's own solver test."""
import warnings

import numpy as np
import pytest
import scipy.sparse

pytestmark = pytest.mark.gpu


def test_cg_solver_kat_seed101():
    """reference tests/test_solver_lib.py:33-43: CG on the seed-101 matrix (6 decimals) and the
    UserWarning when maxiter = 1."""
    import vittles_amd as va
    np.random.seed(101)
    d = 10
    h = np.random.random((d, d))
    h = h + h.T + d * np.eye(d)
    hs = scipy.sparse.csc_matrix(h)
    v = np.random.random(d)
    ref = np.linalg.solve(h, v)
    np.testing.assert_array_almost_equal(va.solver_lib.get_cg_solver(lambda x: h @ x, d)(v), ref)
    np.testing.assert_array_almost_equal(va.solver_lib.get_cg_solver(lambda x: hs @ x, d)(v), ref)
    with pytest.warns(UserWarning, match="CG exited with error code 1"):
        out = va.solver_lib.get_cg_solver(lambda x: hs @ x, d, cg_opts={'maxiter': 1})(v)
    assert out.shape == (d,)            # x is returned regardless (solver_lib.py:97)


@pytest.mark.parametrize("N,D", [(3000, 50), (2000, 400), (300, 3000)])
def test_cg_registered_hvp_matches_oracle(N, D, relerr):
    """Device CG with the fused HVP kernel == oracle.cg_legacy with the NumPy HVP, same tol."""
    import torch
    import vittles_amd as va
    from vittles_amd.cg import GLMHessianVectorProduct, cg_solve
    from oracle import glm, solvers
    X, y, w, _ = glm.synth(N, D, "logistic", seed=N + D)
    rng = np.random.default_rng(1)
    theta = rng.standard_normal(D) / np.sqrt(D)
    prior = 2.0
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior)
    hvp = GLMHessianVectorProduct(obj, theta, w)
    b = rng.standard_normal(D)
    v = rng.standard_normal(D)
    assert relerr(hvp(v), glm.hvp("logistic", X, y, theta, w, v, prior)) < 1e-10
    mv = lambda u: glm.hvp("logistic", X, y, theta, w, u, prior)
    for tol in (1e-5, 1e-12):
        x_ref, info_ref, it_ref = solvers.cg_legacy(mv, b, tol=tol, return_iters=True)
        solve = va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': tol})
        x = solve(b)
        # same iteration, different reduction order: iterates agree far below the stopping tolerance
        assert relerr(x, x_ref) < max(1e-10, 50 * tol * 1e-3), (tol, relerr(x, x_ref))
        xd, info, iters = cg_solve(lambda p, out: hvp.apply(p, out), torch.tensor(b, device="cuda"), tol=tol)
        assert info == info_ref == 0 and abs(iters - it_ref) <= 1
    H = glm.hessian("logistic", X, y, theta, w, prior)
    assert relerr(va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': 1e-13})(b), np.linalg.solve(H, b)) < 1e-10


def test_cg_config4_d20000(relerr):
    """BASELINE.json configs[3] at its full width: D = 20 000 ridge-logistic, matrix-free CG through the
    wide-row fused HVP kernel (k_rows_wide), against the oracle's restatement of the same iteration over
    a NumPy HVP (N kept at a few thousand rows so the oracle finishes in seconds)."""
    import torch
    import vittles_amd as va
    from vittles_amd.cg import GLMHessianVectorProduct, cg_solve
    from oracle import glm, solvers
    N, D = 3000, 20000
    X, y, w, _ = glm.synth(N, D, "logistic", seed=20)
    rng = np.random.default_rng(2)
    theta = rng.standard_normal(D) / np.sqrt(D)
    w = rng.uniform(0.5, 1.5, N)
    prior = 1.0
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior)
    hvp = GLMHessianVectorProduct(obj, theta, w)
    _, l1, l2 = glm.link("logistic", X @ theta, y)
    dvec = w * l2
    mv = lambda u: X.T @ (dvec * (X @ u)) + prior * u            # NumPy HVP (B3's mat_times_vec)
    v = rng.standard_normal(D)
    assert relerr(hvp(v), mv(v)) < 1e-10
    b = X.T @ (l1 * rng.standard_normal(N))                       # jac0 @ dw: a directional sensitivity solve
    bd = torch.tensor(b, device="cuda")
    for tol in (1e-5, 1e-12):
        x_ref, info_ref, it_ref = solvers.cg_legacy(mv, b, tol=tol, return_iters=True)
        xd, info, iters = cg_solve(lambda p, out: hvp.apply(p, out), bd, tol=tol)
        assert info == info_ref == 0 and abs(iters - it_ref) <= 1, (tol, iters, it_ref)
        assert relerr(xd.cpu().numpy(), x_ref) < max(1e-10, 50 * tol * 1e-3), (tol, relerr(xd.cpu().numpy(), x_ref))
        x = va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': tol})(b)
        assert relerr(x, x_ref) < max(1e-10, 50 * tol * 1e-3)
        # true residual of the device solution, by the oracle's HVP
        assert np.linalg.norm(mv(x) - b) <= max(2 * tol, 1e-10) * np.linalg.norm(b)


def test_cg_directional_sensitivity(relerr):
    """The CG configuration is a directional solve: d theta = -H^{-1} (jac0 @ dw)."""
    import torch
    import vittles_amd as va
    from vittles_amd.cg import GLMHessianVectorProduct
    from vittles_amd import rows
    from oracle import glm
    N, D = 4000, 60
    X, y, w, _ = glm.synth(N, D, "logistic", seed=4)
    theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    hvp = GLMHessianVectorProduct(obj, theta, w)
    dw = np.random.default_rng(0).standard_normal(N)
    r = obj.fused(theta, w)["r"]
    rhs = rows.gemv_t(obj.X, r * torch.tensor(dw, device="cuda"))      # jac0 @ dw = X^T (r .* dw)
    dtheta = -1 * va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': 1e-13})(rhs)
    _, _, S = glm.sensitivity("logistic", X, y, theta, w, prior=1.0)
    assert relerr(dtheta.cpu().numpy(), S @ dw) < 1e-9
