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


# ---- multi-vector HVP / block CG (build-side extension) and the cg_opts pass-through --------
@pytest.mark.parametrize("N,D,K", [(3000, 50, 4), (2500, 400, 7), (700, 3001, 16), (5000, 129, 33), (130, 2000, 128),
                                   (4096, 1000, 1)])
def test_glm_hvp_multi_vs_oracle(N, D, K, relerr):
    """vt_glm_hvp_multi: K products from two MFMA passes over X == K NumPy HVPs (1e-10)."""
    import torch
    import vittles_amd as va
    from vittles_amd import rows
    from oracle import glm
    X, y, w, _ = glm.synth(N, D, "logistic", seed=N + D + K)
    rng = np.random.default_rng(K)
    theta = rng.standard_normal(D) / np.sqrt(D)
    prior = np.linspace(0.5, 2.0, D)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior)
    _, _, d = rows.glm_link(obj, theta, torch.tensor(w, device="cuda"))
    V = rng.standard_normal((K, D))
    got = rows.glm_hvp_multi(obj.X, d, torch.tensor(V, device="cuda"), prior=obj.prior).cpu().numpy()
    ref = np.stack([glm.hvp("logistic", X, y, theta, w, V[k], prior) for k in range(K)])
    assert relerr(got, ref) < 1e-10
    # a strided V (columns of a wider matrix) and the scalar prior
    Vw = torch.tensor(np.hstack([V, V]), device="cuda")[:, :D]
    got2 = rows.glm_hvp_multi(obj.X, d, Vw, prior_scalar=0.0).cpu().numpy()
    assert relerr(got2, ref - prior[None, :] * V) < 1e-10


@pytest.mark.parametrize("N,D,K", [(3000, 50, 5), (2000, 400, 12), (300, 3000, 3)])
def test_block_cg_matches_k_independent_oracle_runs(N, D, K, relerr):
    """get_cg_solver(...)(V) with a (D, K) V: every column equals an independent oracle.cg_legacy run
    (vittles/solver_lib.py:91-97 applied column by column under sensitivity_lib.py:226)."""
    import vittles_amd as va
    from vittles_amd.cg import GLMHessianVectorProduct
    from oracle import glm, solvers
    X, y, w, _ = glm.synth(N, D, "logistic", seed=N + D)
    rng = np.random.default_rng(2)
    theta = rng.standard_normal(D) / np.sqrt(D)
    prior = 1.5
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior)
    hvp = GLMHessianVectorProduct(obj, theta, w)
    mv = lambda u: glm.hvp("logistic", X, y, theta, w, u, prior)
    B = rng.standard_normal((D, K)) * np.logspace(0, 3, K)[None, :]     # columns of very different scale
    B[:, 1] = 0.0                                                        # a zero right-hand side converges at once
    for tol in (1e-5, 1e-11):
        Xs = va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': tol})(B)
        assert Xs.shape == (D, K)
        for k in range(K):
            x_ref, info_ref = solvers.cg_legacy(mv, B[:, k], tol=tol)
            assert info_ref == 0
            if np.any(B[:, k]):
                assert relerr(Xs[:, k], x_ref) < max(1e-9, 50 * tol * 1e-3), (tol, k)
            else:
                assert not Xs[:, k].any()
    H = glm.hessian("logistic", X, y, theta, w, prior)
    assert relerr(va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': 1e-13})(B), np.linalg.solve(H, B)) < 1e-9
    # a plain user callable (host matvec) takes the same block path, column by column
    Xu = va.solver_lib.get_cg_solver(mv, D, cg_opts={'tol': 1e-12})(B[:, :3])
    assert relerr(Xu, np.linalg.solve(H, B[:, :3])) < 1e-9


def test_cg_opts_preconditioner_and_callback(relerr):
    """cg_opts['M'] (diagonal) and ['callback'] go through as in scipy.sparse.linalg.cg (solver_lib.py:93)."""
    import scipy.sparse.linalg as spla
    import vittles_amd as va
    rng = np.random.default_rng(7)
    d = 300
    A = rng.standard_normal((d, d)) / np.sqrt(d)
    h = A @ A.T + np.diag(np.logspace(0, 4, d))           # badly scaled: the Jacobi preconditioner matters
    b = rng.standard_normal(d)
    minv = 1.0 / np.diag(h)
    seen_ref, seen = [], []
    x_ref, info_ref = spla.cg(h, b, rtol=1e-10, atol=0.0, M=scipy.sparse.diags(minv), callback=lambda xk: seen_ref.append(xk.copy()))
    assert info_ref == 0
    for M in (minv, np.diag(minv), scipy.sparse.diags(minv).tocsc()):
        seen.clear()
        solve = va.solver_lib.get_cg_solver(lambda v: h @ v, d, cg_opts={'tol': 1e-10, 'M': M, 'callback': lambda xk: seen.append(xk.copy())})
        x = solve(b)
        assert relerr(x, x_ref) < 1e-8 and relerr(x, np.linalg.solve(h, b)) < 1e-8
        assert abs(len(seen) - len(seen_ref)) <= 1 and isinstance(seen[0], np.ndarray)
        assert relerr(seen[3], seen_ref[3]) < 1e-8           # same iterates, not only the same limit
    n_plain = []
    va.solver_lib.get_cg_solver(lambda v: h @ v, d, cg_opts={'tol': 1e-10, 'callback': lambda xk: n_plain.append(1)})(b)
    assert len(n_plain) > 2 * len(seen)                      # the preconditioner did its job
    with pytest.raises(NotImplementedError, match="DIAGONAL"):
        va.solver_lib.get_cg_solver(lambda v: h @ v, d, cg_opts={'M': h})(b)
    with pytest.raises(TypeError, match="unsupported cg_opts"):
        va.solver_lib.get_cg_solver(lambda v: h @ v, d, cg_opts={'bogus': 1})
