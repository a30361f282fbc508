"""GPU parity of the batched re-weighting row: the reference's
``predict_opt_par_from_hyper_par`` (vittles/sensitivity_lib.py:236-247) applied to MANY weight
vectors at once (bootstrap draws as in docs/source/example_notebooks/example_utils.py:19-20,
leave-one-out / K-fold weights as in the notebook's cells 24-26).  Checked against the oracle's
restatement of the single-vector method, column by column.  Tolerance 1e-10 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-10


@pytest.mark.parametrize("N,Da,Db", [(1, 1, 1), (5, 3, 2), (1000, 7, 20), (4099, 64, 100), (3000, 100, 64),
                                     (2049, 130, 100), (777, 128, 257), (5001, 33, 9), (4096, 16, 1000),
                                     (100000, 200, 100), (1500, 1500, 1400)])   # last: 132 tiles, no XCD split
@pytest.mark.parametrize("scaled", [False, True])
def test_gemm_tn(N, Da, Db, scaled, relerr):
    import torch
    from vittles_amd import rows
    rng = np.random.default_rng(N + Da + Db)
    A = rng.standard_normal((N, Da))
    B = rng.standard_normal((N, Db))
    s = rng.standard_normal(N) if scaled else None
    bias = rng.standard_normal(Db) if scaled else None
    alpha = -0.75 if scaled else 1.0
    ref = alpha * (A.T @ (B if s is None else s[:, None] * B))
    if bias is not None:
        ref = ref + bias[None, :]
    t = lambda a: None if a is None else torch.tensor(a, device="cuda")
    out = rows.gemm_tn(t(A), t(B), rowscale=t(s), alpha=alpha, bias=t(bias))
    assert relerr(out.cpu().numpy(), ref) < TOL


def test_gemm_tn_strided_operands(relerr):
    # operands that are column slices of wider matrices (odd leading dimensions, unaligned bases):
    # the direct-from-global kernel
    import torch
    from vittles_amd import rows
    rng = np.random.default_rng(5)
    Abig = torch.tensor(rng.standard_normal((3001, 37)), device="cuda")
    Bbig = torch.tensor(rng.standard_normal((3001, 51)), device="cuda")
    A, B = Abig[:, 3:20], Bbig[:, 1:50]
    out = rows.gemm_tn(A, B)
    assert relerr(out.cpu().numpy(), A.cpu().numpy().T @ B.cpu().numpy()) < TOL


@pytest.mark.parametrize("N,D,K", [(500, 9, 5), (4096, 100, 64), (20000, 20, 300)])
@pytest.mark.parametrize("container", ["numpy", "torch"])
def test_predict_batch_matches_single_vector_predict(N, D, K, container, relerr):
    import torch
    import vittles_amd as va
    from oracle import glm
    from oracle.sensitivity import HyperparameterSensitivityLinearApproximation as OracleLA
    X, y, w, _ = glm.synth(N, D, "logistic", seed=N + K)
    theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)
    rng = np.random.default_rng(K)
    # bootstrap-style weights (multinomial counts) next to leave-one-out indicator weights
    W = rng.multinomial(N, np.ones(N) / N, size=K).T.astype(np.float64)
    for k in range(min(K, 3)):
        W[:, k] = 1.0
        W[k, k] = 0.0
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    if container == "torch":
        la = va.HyperparameterSensitivityLinearApproximation(
            obj, torch.tensor(theta, device="cuda"), torch.tensor(w, device="cuda"))
        got = la.predict_opt_par_from_hyper_par_batch(torch.tensor(W, device="cuda"))
        assert isinstance(got, torch.Tensor) and got.is_cuda
        got = got.cpu().numpy()
    else:
        la = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
        got = la.predict_opt_par_from_hyper_par_batch(W)
        assert isinstance(got, np.ndarray)
    assert got.shape == (D, K)
    ref_la = OracleLA(
        grad=lambda t, ww: glm.grad("logistic", X, y, t, ww, prior=1.0),
        hess=lambda t, ww: glm.hessian("logistic", X, y, t, ww, prior=1.0),
        cross_hess=lambda t, ww: glm.cross_hessian("logistic", X, y, t, ww),
        opt_par_value=theta, hyper_par_value=w)
    ref = np.stack([ref_la.predict_opt_par_from_hyper_par(W[:, k]) for k in range(K)], axis=1)
    assert relerr(got, ref) < TOL
    # and the build's own single-vector method, column by column
    if container == "torch":
        one = la.predict_opt_par_from_hyper_par(torch.tensor(W[:, 1], device="cuda")).cpu().numpy()
    else:
        one = la.predict_opt_par_from_hyper_par(W[:, 1])
    assert relerr(got[:, 1], one) < TOL
    with pytest.raises(ValueError):
        la.predict_opt_par_from_hyper_par_batch(W[:-1] if container == "numpy"
                                                else torch.tensor(W[:-1], device="cuda"))
