"""GPU parity of the large-D (D > 128) dense path: tiled SYRK, blocked Cholesky, blocked MFMA TRSM
(BASELINE.json configs[2] shape class, D = 1000) against the oracle."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize("D", [129, 200, 256, 333, 1000])
def test_potrf_potrs_large(D, relerr):
    import torch
    from vittles_amd.linalg import DeviceCholesky
    rng = np.random.default_rng(D)
    A = rng.standard_normal((2 * D, D))
    H = A.T @ A / D + np.eye(D)
    chol = DeviceCholesky.factor(torch.tensor(H, device="cuda"), pack=(D % 2 == 0))
    L = np.tril(chol.L.cpu().numpy())
    assert relerr(L, np.linalg.cholesky(H)) < TOL
    B = rng.standard_normal((D, 300))
    ref = scipy.linalg.cho_solve(scipy.linalg.cho_factor(H), B)
    got = chol.solve(torch.tensor(B, device="cuda"))
    assert got.shape == (D, 300) and relerr(got.cpu().numpy(), ref) < 1e-9
    v = rng.standard_normal(D)
    assert relerr(chol.solve(torch.tensor(v, device="cuda")).cpu().numpy(), np.linalg.solve(H, v)) < 1e-9
    # identity: exact pass-through of an asymmetric right-hand side
    I = DeviceCholesky.factor(torch.eye(D, dtype=torch.float64, device="cuda"))
    Bm = torch.arange(70 * D, dtype=torch.float64, device="cuda").reshape(70, D) + 1.0
    assert torch.equal(I.solve_rows(Bm), Bm)


def test_potrf_large_not_positive_definite():
    import torch
    from vittles_amd.linalg import DeviceCholesky
    H = np.eye(300)
    H[200, 200] = -1.0
    with pytest.raises(np.linalg.LinAlgError) as e:
        DeviceCholesky.factor(torch.tensor(H, device="cuda"))
    assert "201-th leading minor" in str(e.value)


@pytest.mark.parametrize("N,D,family", [(3000, 130, "logistic"), (5000, 257, "gaussian"), (20000, 1000, "logistic"),
                                         (1501, 1001, "poisson"), (4200, 301, "logistic"),
                                         (2600, 2100, "logistic")])      # 153 tiles: the plan without the XCD split
def test_sensitivity_large_d(N, D, family, relerr):
    from oracle import glm
    import vittles_amd as va
    X, y, w, _ = glm.synth(N, D, family, seed=N + D)
    rng = np.random.default_rng(1)
    w = 0.5 + rng.random(N)
    prior = 1.0
    theta = glm.newton_optimum(family, X, y, w, prior=prior)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior)
    out = obj.fused(theta, w)
    H_ref, jac0, S_ref = glm.sensitivity(family, X, y, theta, w, prior=prior)
    H = out["H"].cpu().numpy()
    assert relerr(H, H_ref) < TOL and np.array_equal(H, H.T)
    assert relerr(out["g"].cpu().numpy() + 1.0, glm.grad(family, X, y, theta, w, prior=prior) + 1.0) < TOL
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, w, validate_optimum=True, grad_tol=1e-7)
    assert relerr(sens.get_dopt_dhyper(), S_ref) < TOL
    w1 = w * (1.0 + 0.01 * np.cos(np.arange(N)))
    assert relerr(sens.predict_opt_par_from_hyper_par(w1), theta + S_ref @ (w1 - w)) < TOL


# ---- mid range 129 <= D <= 256: the one-pass kernels of csrc/vt_hess_mid.hip / vt_trsm_mid.hip ----------------
# every block count NB = 9..16, every strip count (D mod 16 in 1..4, 5..8, 9..15, 0), ragged and short tiles
@pytest.mark.parametrize("N,D", [(1, 129), (23, 144), (24, 160), (4096, 129), (4111, 133), (5000, 137), (4099, 144),
                                 (6001, 150), (4100, 160), (4200, 161), (4500, 176), (4800, 177), (5001, 192),
                                 (4097, 200), (4300, 208), (4444, 215), (4600, 224), (4700, 231), (4900, 240),
                                 (5100, 249), (8000, 255), (7777, 256),
                                 # 256 < D <= 384: one-pass kernel on the first 256 columns + strip product (round 3)
                                 (1, 257), (23, 260), (4096, 257), (4111, 273), (5000, 300), (4099, 320), (4200, 337),
                                 (4300, 352), (4100, 353), (4400, 369), (4500, 384), (4100, 385)])
@pytest.mark.parametrize("family", ["logistic", "poisson", "gaussian"])
def test_fused_hessian_grad_mid(N, D, family, relerr):
    from oracle import glm
    import vittles_amd as va
    X, y, w, _ = glm.synth(N, D, family, N + D)
    rng = np.random.default_rng(N + D + 100)
    w = 0.5 + rng.random(N)
    theta = rng.standard_normal(D) / np.sqrt(D)
    offset = 0.1 * rng.standard_normal(N) if D % 2 == 0 else None
    prior = np.linspace(0.5, 2.0, D)
    scale = 1.5
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior, scale=scale, offset=offset)
    assert obj.X.stride(0) % 2 == 0 and obj.X.data_ptr() % 16 == 0
    out = obj.fused(theta, w)
    H = out["H"].cpu().numpy()
    assert relerr(H, glm.hessian(family, X, y, theta, w, prior, scale, offset)) < TOL
    assert np.array_equal(H, H.T), "H must be exactly symmetric"
    assert relerr(out["g"].cpu().numpy(), glm.grad(family, X, y, theta, w, prior, scale, offset)) < TOL
    _, l1, l2 = glm.link(family, glm.linear_predictor(X, theta, offset), y)
    assert relerr(out["r"].cpu().numpy(), scale * l1) < TOL
    assert relerr(out["d"].cpu().numpy(), scale * w * l2) < TOL


@pytest.mark.parametrize("N,D,family", [(4096, 129, "logistic"), (5000, 160, "poisson"), (4321, 192, "gaussian"),
                                         (6000, 255, "logistic"), (4200, 256, "logistic"), (4100, 145, "gaussian"),
                                         (4500, 216, "poisson"), (4300, 273, "logistic"), (5000, 320, "poisson"),
                                         (4100, 322, "gaussian")])
def test_sensitivity_mid_d(N, D, family, relerr):
    """D in {129, 160, 192, 255} x N >= 4096 through the drop-in class vs the oracle."""
    from oracle import glm
    import vittles_amd as va
    X, y, w, _ = glm.synth(N, D, family, seed=N + D)
    rng = np.random.default_rng(2)
    w = 0.5 + rng.random(N)
    theta = glm.newton_optimum(family, X, y, w, prior=1.0)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=1.0)
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, w, validate_optimum=True, grad_tol=1e-7)
    H_ref, _, S_ref = glm.sensitivity(family, X, y, theta, w, prior=1.0)
    assert relerr(sens.get_hessian_at_opt(), H_ref) < TOL
    assert relerr(sens.get_dopt_dhyper(), S_ref) < TOL


# the chain kernels of csrc/vt_trsm_mid.h directly: every block count NB = 9..24 with every last-block class
# (1..4, 5..8, 9..15, 16 columns), one / ragged / many tiles, odd row pitches, in place; past 384 columns the blocked
# solve on 384-column diagonal blocks (forward-only / backward-only halves of the NB = 24 chain + GEMM-NT updates)
@pytest.mark.parametrize("D", [129, 133, 137, 144, 145, 150, 158, 160, 161, 166, 171, 176, 178, 185, 190, 192, 193,
                               200, 207, 208, 212, 216, 223, 224, 228, 231, 239, 240, 241, 247, 255, 256,
                               257, 261, 266, 272, 273, 280, 284, 288, 289, 296, 300, 304, 305, 311, 316, 320,
                               321, 330, 336, 337, 345, 352, 360, 368, 369, 380, 384,
                               385, 400, 449, 640, 641, 700, 768, 769, 800])
def test_trsm_rows_mid(D, relerr):
    import torch
    from vittles_amd.linalg import DeviceCholesky
    rng = np.random.default_rng(D)
    A = rng.standard_normal((2 * D, D))
    H = A.T @ A / D + np.eye(D)
    cf = scipy.linalg.cho_factor(H)
    chol = DeviceCholesky.factor(torch.tensor(H, device="cuda"), pack=True)
    for N, pad in [(1, 0), (17, 1), (4096 + D, 0), (40000 + D if D <= 384 else 9000 + D, 3)]:
        B = rng.standard_normal((N, D + pad))
        s = 0.5 + rng.random(N)
        Bd = torch.tensor(B, device="cuda")[:, :D]
        out = torch.full((N, D + pad), -7.0, dtype=torch.float64, device="cuda")
        got = chol.solve_rows(Bd, rowscale=torch.tensor(s, device="cuda"), alpha=-1.5, out=out[:, :D])
        ref = -1.5 * s[:, None] * scipy.linalg.cho_solve(cf, B[:, :D].T).T
        assert relerr(got.cpu().numpy(), ref) < 1e-9
        assert torch.all(out[:, D:] == -7.0), "columns past D must not be written"
        Bi = Bd.clone() if pad == 0 else Bd
        chol.solve_rows(Bi, out=Bi)     # in place, no row scale
        assert relerr(Bi.cpu().numpy(), scipy.linalg.cho_solve(cf, B[:, :D].T).T) < 1e-9
