"""GPU parity (through the C-ABI / drop-in classes) of the weighted-GLM data-weight
sensitivity pipeline against the CPU oracle.  Tolerance: 1e-10 relative
(max|a-ref|/max|ref| per array), the fp64 bar BASELINE.json's north_star states."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _problem(N, D, family, seed, prior=1.0, with_offset=False):
    from oracle import glm
    X, y, w, _ = glm.synth(N, D, family, seed)
    rng = np.random.default_rng(seed + 100)
    w = 0.5 + rng.random(N)
    theta = rng.standard_normal(D) / np.sqrt(D)
    offset = 0.1 * rng.standard_normal(N) if with_offset else None
    return X, y, w, theta, offset


@pytest.mark.parametrize("N,D", [(1, 1), (3, 5), (17, 16), (1000, 20), (257, 31), (4099, 32),
                                 (2000, 100), (513, 128), (1001, 97), (300, 113),
                                 # N >= 4096: the LDS-tile kernel (48-row tiles, ragged last tile, NB = 4..7)
                                 (4096, 50), (4099, 64), (5003, 77), (4801, 96), (20001, 100), (12289, 112),
                                 (4100, 99), (5000, 128), (4500, 113), (6001, 127), (4096, 120)])
@pytest.mark.parametrize("family", ["logistic", "poisson", "gaussian"])
def test_fused_hessian_grad(N, D, family, relerr):
    from oracle import glm
    import vittles_amd as va
    X, y, w, theta, offset = _problem(N, D, family, seed=N + D, with_offset=(D % 2 == 0))
    prior = np.linspace(0.5, 2.0, D)
    scale = 1.5
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior, scale=scale, offset=offset)
    if D > 80:   # odd D: rows re-laid for the 16-byte DMA lanes of the tile kernel
        assert obj.X.stride(0) % 2 == 0 and obj.X.data_ptr() % 16 == 0
    out = obj.fused(theta, w)
    H = out["H"].cpu().numpy()
    assert relerr(H, glm.hessian(family, X, y, theta, w, prior, scale, offset)) < TOL
    assert np.array_equal(H, H.T), "H must be exactly symmetric"
    assert relerr(out["g"].cpu().numpy(), glm.grad(family, X, y, theta, w, prior, scale, offset)) < TOL
    _, l1, l2 = glm.link(family, glm.linear_predictor(X, theta, offset), y)
    assert relerr(out["r"].cpu().numpy(), scale * l1) < TOL
    assert relerr(out["d"].cpu().numpy(), scale * w * l2) < TOL


@pytest.mark.parametrize("D", [1, 2, 7, 16, 20, 33, 64, 100, 127, 128])
def test_potrf_potrs(D, relerr):
    torch = _torch()
    from vittles_amd.linalg import DeviceCholesky
    rng = np.random.default_rng(D)
    A = rng.standard_normal((D + 5, D))
    H = A.T @ A + np.eye(D)
    chol = DeviceCholesky.factor(torch.tensor(H, device="cuda"))
    L = np.tril(chol.L.cpu().numpy())
    assert relerr(L, np.linalg.cholesky(H)) < TOL
    # vector and matrix right-hand sides through the generic kernel
    v = rng.standard_normal(D)
    ref = scipy.linalg.cho_solve(scipy.linalg.cho_factor(H), v)
    assert relerr(chol.solve(torch.tensor(v, device="cuda")).cpu().numpy(), ref) < 1e-9
    B = rng.standard_normal((D, 7))
    refB = scipy.linalg.cho_solve(scipy.linalg.cho_factor(H), B)
    assert relerr(chol.solve(torch.tensor(B, device="cuda")).cpu().numpy(), refB) < 1e-9
    # MFMA TRSM path (>= 64 right-hand sides)
    B = rng.standard_normal((D, 333))
    refB = scipy.linalg.cho_solve(scipy.linalg.cho_factor(H), B)
    got = chol.solve(torch.tensor(B, device="cuda"))
    assert got.shape == (D, 333)
    assert relerr(got.cpu().numpy(), refB) < 1e-9


def test_potrf_not_positive_definite():
    torch = _torch()
    from vittles_amd.linalg import DeviceCholesky
    H = np.eye(6)
    H[3, 3] = -1.0
    with pytest.raises(np.linalg.LinAlgError) as e:
        DeviceCholesky.factor(torch.tensor(H, device="cuda"))
    assert "4-th leading minor" in str(e.value)   # same text as scipy.linalg.cho_factor


def test_trsm_rows_identity_layout():
    """A = I-type check with an ASYMMETRIC right-hand side: with H = I the solve must
    return B itself, element for element (catches any row/column swap in the MFMA
    operand maps)."""
    torch = _torch()
    from vittles_amd.linalg import DeviceCholesky
    for D in (16, 20, 100, 128):
        chol = DeviceCholesky.factor(torch.eye(D, dtype=torch.float64, device="cuda"))
        B = torch.arange(70 * D, dtype=torch.float64, device="cuda").reshape(70, D) + 1.0
        out = chol.solve_rows(B)
        assert torch.equal(out, B), D


@pytest.mark.parametrize("N,D", [(1, 3), (16, 16), (33, 20), (1000, 20), (4099, 100), (777, 128), (500, 45)])
@pytest.mark.parametrize("family", ["logistic", "gaussian"])
def test_sensitivity_end_to_end(N, D, family, relerr):
    """HyperparameterSensitivityLinearApproximation on a registered objective ==
    oracle's -cho_solve(cho_factor(H), jac0)  (reference sensitivity_lib.py:226)."""
    from oracle import glm
    import vittles_amd as va
    X, y, w, _, _ = _problem(N, D, family, seed=7 * N + D)
    prior = 1.0
    theta = glm.newton_optimum(family, X, y, w, prior=prior)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior)
    sens = va.HyperparameterSensitivityLinearApproximation(
        obj, theta, w, validate_optimum=True, grad_tol=1e-8)
    H, jac0, S = glm.sensitivity(family, X, y, theta, w, prior=prior)
    got = sens.get_dopt_dhyper()
    assert got.shape == (D, N) and got.flags.f_contiguous
    assert relerr(got, S) < TOL
    assert relerr(sens.get_hessian_at_opt(), H) < TOL
    # predict (K9): theta + S (w1 - w0)
    w1 = w * (1.0 + 0.01 * np.cos(np.arange(N)))
    assert relerr(sens.predict_opt_par_from_hyper_par(w1), theta + S @ (w1 - w)) < TOL


def test_sensitivity_torch_in_torch_out(relerr):
    torch = _torch()
    from oracle import glm
    import vittles_amd as va
    N, D = 3000, 100
    X, y, w, _, _ = _problem(N, D, "logistic", seed=3)
    theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)
    Xd, yd, wd, td = (torch.tensor(a, device="cuda") for a in (X, y, w, theta))
    obj = va.WeightedGLMObjective(Xd, yd, family="logistic", prior=1.0)
    sens = va.HyperparameterSensitivityLinearApproximation(obj, td, wd)
    got = sens.get_dopt_dhyper()
    assert isinstance(got, torch.Tensor) and got.is_cuda and got.shape == (D, N)
    _, _, S = glm.sensitivity("logistic", X, y, theta, w, prior=1.0)
    assert relerr(got.cpu().numpy(), S) < TOL


def test_validate_optimum_raises():
    import vittles_amd as va
    X, y, w, theta, _ = _problem(200, 8, "logistic", seed=1)
    obj = va.WeightedGLMObjective(X, y, prior=1.0)
    with pytest.raises(ValueError) as e:
        va.HyperparameterSensitivityLinearApproximation(obj, theta + 1.0, w, validate_optimum=True)
    assert "The estimating equation is not zero at the proposed" in str(e.value)


def test_row_kernels(relerr):
    """K0 link, K3 gradient, K6 HVP, K9 gemv_t, value -- against the oracle."""
    torch = _torch()
    from oracle import glm
    import vittles_amd as va
    from vittles_amd import rows
    for (N, D) in [(5, 3), (1000, 20), (3001, 100), (700, 257), (300, 1000), (70, 2500), (40, 20000), (33, 4097),
                   (600, 5000), (520, 12001), (300, 15000), (530, 20000), (270, 20448), (513, 4099), (260, 20447)]:
        X, y, w, theta, offset = _problem(N, D, "logistic", seed=N + D, with_offset=True)
        prior = np.linspace(0.5, 1.5, D)
        obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior, scale=0.7, offset=offset)
        g, r, d = rows.glm_grad(obj, theta, w, prior=obj.prior)
        assert relerr(g.cpu().numpy(), glm.grad("logistic", X, y, theta, w, prior, 0.7, offset)) < TOL, (N, D)
        z, r2, d2 = rows.glm_link(obj, theta, w, want_z=True)
        assert relerr(z.cpu().numpy(), glm.linear_predictor(X, theta, offset)) < TOL
        assert torch.equal(r, r2) and torch.equal(d, d2)
        v = np.random.default_rng(0).standard_normal(D)
        hv = rows.glm_hvp(obj.X, d, torch.tensor(v, device="cuda"), prior=obj.prior)
        assert relerr(hv.cpu().numpy(), glm.hvp("logistic", X, y, theta, w, v, prior, 0.7, offset)) < TOL, (N, D)
        u = np.random.default_rng(1).standard_normal(N)
        assert relerr(rows.gemv_t(obj.X, torch.tensor(u, device="cuda")).cpu().numpy(), X.T @ u) < TOL
        assert relerr(rows.gemv(obj.X, torch.tensor(v, device="cuda")).cpu().numpy(), X @ v) < TOL
        val = obj(theta, w)
        ref = glm.value("logistic", X, y, theta, w, prior, 0.7, offset)
        assert abs(val - ref) <= TOL * abs(ref), (N, D)


@pytest.mark.parametrize("D,K", [(1, 1), (20, 1), (100, 1), (100, 5), (128, 3), (300, 2)])
def test_vt_potrs_generic_kernel_direct(D, K, relerr):
    """The generic substitution kernel through the C-ABI (DeviceCholesky.solve routes small D to the
    MFMA tile kernel, so it is called directly here)."""
    torch = _torch()
    from vittles_amd import _lib, device as dev
    from vittles_amd.linalg import DeviceCholesky
    rng = np.random.default_rng(D + K)
    A = rng.standard_normal((D + 5, D))
    H = A.T @ A + np.eye(D)
    chol = DeviceCholesky.factor(torch.tensor(H, device="cuda"))
    B = rng.standard_normal((D, K))
    x = torch.tensor(B, device="cuda").contiguous()
    _lib.check(_lib.load().vt_potrs(chol.L.data_ptr(), D, D, x.data_ptr(), K, 1, K, dev.stream_ptr()), "vt_potrs")
    assert relerr(x.cpu().numpy(), np.linalg.solve(H, B)) < 1e-9


@pytest.mark.parametrize("D", [5, 100, 200])
def test_empty_data_set(D, relerr):
    """N = 0 rows (an empty hyper-parameter vector): H = prior, g = prior .* theta, S has shape (D, 0),
    predictions return theta-hat -- what autograd + SciPy give the reference for an empty sum."""
    torch = _torch()
    import vittles_amd as va
    from vittles_amd import rows
    X = np.zeros((0, D)); y = np.zeros(0); w = np.zeros(0)
    theta = 0.1 * np.arange(D)
    prior = np.linspace(1.0, 2.0, D)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior)
    out = obj.fused(theta, w)
    assert relerr(out["H"].cpu().numpy(), np.diag(prior)) == 0.0
    assert relerr(out["g"].cpu().numpy(), prior * theta) < 1e-15
    assert abs(obj(theta, w) - 0.5 * np.sum(prior * theta ** 2)) < 1e-12
    la = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
    assert la.get_dopt_dhyper().shape == (D, 0)
    assert relerr(la.predict_opt_par_from_hyper_par(w), theta) == 0.0
    z = lambda *s: torch.zeros(s, dtype=torch.float64, device="cuda")
    bias = torch.arange(4, dtype=torch.float64, device="cuda")
    assert np.array_equal(rows.gemm_tn(z(0, 3), z(0, 4), bias=bias).cpu().numpy(), np.tile(np.arange(4.0), (3, 1)))
    th, info = va.optim.newton_glm(obj, w)
    assert info["converged"] and np.abs(th).max() < 1e-12          # minimiser of the prior alone


@pytest.mark.parametrize("N,D", [(3, 5), (257, 31), (1001, 97), (300, 113), (5003, 77), (4100, 99), (4500, 113),
                                 (6001, 127), (4099, 65), (700, 257)])
@pytest.mark.parametrize("family", ["logistic", "poisson", "gaussian"])
def test_fused_hessian_grad_unaligned_rows(N, D, family, relerr, monkeypatch):
    """Odd leading dimension kept as it is (no re-layout): the register-streaming kernels, which take any
    8-byte aligned rows, must give the same results as the 16-byte-lane tile kernels."""
    from oracle import glm
    import vittles_amd as va
    monkeypatch.setattr(va.objectives, "RELAYOUT_UNALIGNED_ROWS", False)
    X, y, w, theta, offset = _problem(N, D, family, seed=N + D, with_offset=True)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=1.0, offset=offset)
    assert obj.X.stride(0) == D
    out = obj.fused(theta, w)
    assert relerr(out["H"].cpu().numpy(), glm.hessian(family, X, y, theta, w, 1.0, 1.0, offset)) < TOL
    assert relerr(out["g"].cpu().numpy(), glm.grad(family, X, y, theta, w, 1.0, 1.0, offset)) < TOL
    _, l1, l2 = glm.link(family, glm.linear_predictor(X, theta, offset), y)
    assert relerr(out["r"].cpu().numpy(), l1) < TOL
    assert relerr(out["d"].cpu().numpy(), w * l2) < TOL
    lin = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
    _, _, S = glm.sensitivity(family, X, y, theta, w, 1.0, 1.0, offset)
    assert relerr(np.asarray(lin.get_dopt_dhyper()), S) < TOL * max(1.0, np.linalg.cond(out["H"].cpu().numpy()))


@pytest.mark.parametrize("D", [20, 70])
def test_fused_hessian_huge_leading_dimension(D, relerr, monkeypatch):
    """X as a column slice of a very wide matrix (row stride beyond the 32-bit offsets of the row-link kernel's
    buffer descriptor): the dispatcher must take another kernel, with the same result."""
    torch = _torch()
    from oracle import glm
    import vittles_amd as va
    N, wide = 5, (1 << 22) + 64
    rng = np.random.default_rng(D)
    big = torch.zeros((N, wide), dtype=torch.float64, device="cuda")
    X = rng.standard_normal((N, D))
    big[:, 3:3 + D] = torch.tensor(X, device="cuda")
    y = (rng.random(N) < 0.5).astype(float)
    w = 0.5 + rng.random(N)
    theta = rng.standard_normal(D) / np.sqrt(D)
    monkeypatch.setattr(va.objectives, "RELAYOUT_UNALIGNED_ROWS", False)   # keep the caller's layout
    obj = va.WeightedGLMObjective(big[:, 3:3 + D], y, prior=1.0)
    assert obj.X.stride(0) == wide
    out = obj.fused(theta, w)
    assert relerr(out["H"].cpu().numpy(), glm.hessian("logistic", X, y, theta, w, 1.0)) < TOL
    assert relerr(out["g"].cpu().numpy(), glm.grad("logistic", X, y, theta, w, 1.0)) < TOL
