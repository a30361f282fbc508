"""The counter-based generator (SURVEY.md section 8d "Synthetic inputs"): device kernels vs the NumPy mirror, and
the property the design is for -- any row block regenerates independently of how the rows were sharded."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


@pytest.mark.parametrize("N,D,row0", [(1, 1, 0), (7, 5, 3), (1000, 20, 0), (513, 101, 2 ** 33 + 5), (64, 4096, 17)])
def test_normal_rows_device_matches_host(N, D, row0):
    torch = _torch()
    from vittles_amd import synth
    Xd = synth.normal_rows(11, row0, N, D).cpu().numpy()
    Xh = synth.normal_rows_host(11, row0, N, D)
    # identical Philox bits; log / sincospi of the two libms differ by a few ulps (|x| <= 8.6)
    assert np.abs(Xd - Xh).max() < 2e-14
    X32 = synth.normal_rows(11, row0, N, D, dtype=torch.float32).cpu().numpy()
    assert X32.dtype == np.float32
    assert np.mean(X32 != Xh.astype(np.float32)) < 1e-6      # a float rounding boundary between the two libms


def test_normal_moments_and_streams():
    _torch()
    from vittles_amd import synth
    X = synth.normal_rows(0, 0, 200_000, 10).cpu().numpy()
    assert abs(X.mean()) < 5e-3 and abs(X.std() - 1.0) < 5e-3
    assert abs(np.corrcoef(X[:, 0], X[:, 1])[0, 1]) < 1e-2          # the two Box-Muller outputs of one counter
    assert abs(np.corrcoef(X[:-1, 0], X[1:, 0])[0, 1]) < 1e-2       # consecutive rows
    other_seed = synth.normal_rows(1, 0, 100, 10).cpu().numpy()
    other_stream = synth.normal_rows(0, 0, 100, 10, stream=synth.STREAM_THETA).cpu().numpy()
    assert np.abs(other_seed - X[:100]).min() > 0 and np.abs(other_stream - X[:100]).min() > 0


@pytest.mark.parametrize("family", ["logistic", "poisson", "gaussian"])
def test_glm_problem_shards_are_slices_of_one_data_set(family):
    torch = _torch()
    from vittles_amd import synth
    N, D = 5000, 33
    X, y, th = synth.glm_problem(N, D, family, seed=2)
    # two "ranks" generating their own row ranges reproduce the unsharded data bit for bit
    Xa, ya, tha = synth.glm_problem(2000, D, family, seed=2, row0=0)
    Xb, yb, thb = synth.glm_problem(3000, D, family, seed=2, row0=2000)
    assert torch.equal(torch.cat([Xa, Xb]), X) and torch.equal(torch.cat([ya, yb]), y)
    assert torch.equal(tha, th) and torch.equal(thb, th)
    # host regeneration of arbitrary rows
    idx = np.array([0, 1, 1999, 2000, 4999, 1234])
    Xh, yh = synth.glm_rows_host(N, D, family, 2, idx)
    assert np.abs(X.cpu().numpy()[idx] - Xh).max() < 2e-14
    if family == "gaussian":
        assert np.abs(y.cpu().numpy()[idx] - yh).max() < 1e-12
    else:
        assert np.array_equal(y.cpu().numpy()[idx], yh)
    if family == "logistic":
        z = X.cpu().numpy() @ th.cpu().numpy()
        assert abs(y.mean().item() - (1 / (1 + np.exp(-z))).mean()) < 0.03
    if family == "poisson":
        z = 0.5 * X.cpu().numpy() @ th.cpu().numpy()
        assert abs(y.mean().item() - np.exp(z).mean()) < 0.1


def test_synth_feeds_the_pipeline(relerr):
    """End to end: data generated in HBM, sensitivities on the device, checked on the host from REGENERATED rows."""
    torch = _torch()
    import scipy.linalg
    import vittles_amd as va
    from vittles_amd import synth
    N, D = 20000, 24
    X, y, _ = synth.glm_problem(N, D, "logistic", seed=3)
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    obj = va.WeightedGLMObjective(X, y, prior=1.0)
    theta, info = va.optim.newton_glm(obj, w)
    assert info["converged"]
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
    S = sens.get_dopt_dhyper()                       # CUDA (D, N)
    idx = np.arange(0, N, 997)
    Xh, yh = synth.glm_rows_host(N, D, "logistic", 3, idx)
    th = theta.cpu().numpy()
    p = 1 / (1 + np.exp(-Xh @ th))
    jac = (Xh * (p - yh)[:, None]).T                 # columns n of the cross-Hessian, from regenerated rows
    H = sens.get_hessian_at_opt().cpu().numpy()
    ref = -scipy.linalg.cho_solve(scipy.linalg.cho_factor(H), jac)
    assert relerr(S.cpu().numpy()[:, idx], ref) < 1e-10
