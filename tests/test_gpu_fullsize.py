"""BASELINE.json's configurations at (or past) their FULL sizes through the drop-in class:
  * configs[0] N = 10^4 x D = 20 and configs[1] N = 10^6 x D = 100 against oracle.glm.sensitivity on ALL rows;
  * configs[2]'s shape class, N = 2*10^5 x D = 1000: H against NumPy, sampled rows of S and the all-rows checksum;
  * row counts whose X and S pass 2^32 bytes (D = 8, 20, 100; Philox data that exists only in HBM): sampled rows
    regenerated on the host, the all-rows checksum, and H against the sum of the Hessians of 8 row shards.
The size-independent checks are the scheme of tools/run_full_configs.py: S[n] = -r_n H^{-1} x_n row by row with
SciPy's cho_solve, and sum_n w_n S[n, :] = -H^{-1} (g - prior theta) over every row."""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


@pytest.mark.parametrize("N,D", [(10_000, 20), (1_000_000, 100)])
def test_config_full_size_all_rows_vs_oracle(N, D):
    import torch
    from oracle import glm
    import vittles_amd as va
    X, y, w, _ = glm.synth(N, D, "logistic", seed=0)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    wd = torch.tensor(w, device="cuda")
    theta, info = va.optim.newton_glm(obj, wd, grad_tol=1e-9)
    assert info["converged"]
    th = theta.cpu().numpy()
    assert np.linalg.norm(glm.grad("logistic", X, y, th, w, prior=1.0)) < 1e-6       # the oracle agrees it is the optimum
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, wd, validate_optimum=True, grad_tol=1e-7)
    H_ref, _, S_ref = glm.sensitivity("logistic", X, y, th, w, prior=1.0)              # (D, N), every row
    assert _rel(sens.get_hessian_at_opt().cpu().numpy(), H_ref) < TOL
    S = sens.get_dopt_dhyper()
    assert tuple(S.shape) == (D, N)
    assert _rel(S.cpu().numpy(), S_ref) < TOL
    w1 = w * (1.0 + 0.01 * np.cos(np.arange(N)))
    got = sens.predict_opt_par_from_hyper_par(torch.tensor(w1, device="cuda")).cpu().numpy()
    assert _rel(got, th + S_ref @ (w1 - w)) < TOL


def _device_problem(N, D, seed):
    import torch
    import vittles_amd as va
    X, y, _ = va.synth.glm_problem(N, D, "logistic", seed=seed, device=torch.device("cuda", 0))
    w = 0.75 + 0.5 * torch.rand(N, dtype=torch.float64, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    theta, info = va.optim.newton_glm(obj, w, grad_tol=1e-8)
    assert info["converged"]
    return X, y, w, obj, theta


def _size_independent_checks(N, D, seed, X, y, w, obj, theta, sens, nsample=1500):
    import torch
    import vittles_amd as va
    from vittles_amd import rows
    S = sens.get_dopt_dhyper().t()                      # (N, D) rows, device
    assert tuple(S.shape) == (N, D)
    out = obj.fused(theta, w)
    Hh, gh, th = out["H"].cpu().numpy(), out["g"].cpu().numpy(), theta.cpu().numpy()
    cf = scipy.linalg.cho_factor(Hh)
    # sampled rows, REGENERATED on the host from the counter-based seed (first, last and the rows around 2^32 bytes)
    edge = [0, 1, N - 2, N - 1]
    cross = (1 << 32) // (8 * X.stride(0))
    if cross + 2 < N:
        edge += [cross - 1, cross, cross + 1]
    idx = np.unique(np.concatenate([np.array(edge), np.random.default_rng(seed).integers(0, N, nsample)]))
    Xs, ys = va.synth.glm_rows_host(N, D, "logistic", seed, idx)
    di = torch.as_tensor(idx, device="cuda")
    assert np.abs(X[di].cpu().numpy() - Xs).max() < 1e-12
    assert np.array_equal(y[di].cpu().numpy(), ys)
    p = 1.0 / (1.0 + np.exp(-(Xs @ th)))
    S_ref = -scipy.linalg.cho_solve(cf, (Xs * (p - ys)[:, None]).T).T
    assert _rel(S[di].cpu().numpy(), S_ref) < 1e-9
    # every row: sum_n w_n S[n, :] = -H^{-1} (g - prior theta)
    lhs = rows.gemv_t(S, w).cpu().numpy()
    assert _rel(lhs, -scipy.linalg.cho_solve(cf, gh - th)) < 1e-9
    # H(all rows) = sum of the Hessians of 8 row shards (each well inside 32-bit offsets)
    H_sum = np.zeros((D, D))
    for r in range(8):
        a, b = (N * r) // 8, (N * (r + 1)) // 8
        sh = va.WeightedGLMObjective(X[a:b], y[a:b], family="logistic", prior=1.0 if r == 0 else None)
        H_sum += sh.fused(theta, w[a:b])["H"].cpu().numpy()
    assert _rel(H_sum, Hh) < TOL
    return Hh, th


def test_config3_shape_class_d1000():
    """N = 2*10^5 x D = 1000 (a fiftieth of configs[2]'s rows, its full D)."""
    import vittles_amd as va
    N, D, seed = 200_000, 1000, 3
    X, y, w, obj, theta = _device_problem(N, D, seed)
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, w, validate_optimum=True, grad_tol=1e-6)
    Hh, th = _size_independent_checks(N, D, seed, X, y, w, obj, theta, sens, nsample=800)
    Xh, wh = X.cpu().numpy(), w.cpu().numpy()
    ph = 1.0 / (1.0 + np.exp(-(Xh @ th)))
    assert _rel(Hh, Xh.T @ ((wh * ph * (1 - ph))[:, None] * Xh) + np.eye(D)) < TOL


@pytest.mark.parametrize("N,D", [(4_000_000, 8), (68_000_000, 8), (27_500_000, 20), (5_600_000, 100)])
def test_rows_past_2_to_32_bytes(N, D):
    """X and S of 8 N D bytes each (all but the first case past 2^32): grid / tail behaviour, 64-bit row offsets and the
    persistent-tile loops at row counts no host oracle pass covers."""
    import vittles_amd as va
    seed = 11 + D
    X, y, w, obj, theta = _device_problem(N, D, seed)
    if N > 4_000_000:
        assert 8 * N * X.stride(0) > (1 << 32)
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, w, validate_optimum=True, grad_tol=1e-5)
    _size_independent_checks(N, D, seed, X, y, w, obj, theta, sens)
