"""GPU parity of the optional fp32 STORAGE mode (BASELINE.json north_star: "optional
fp32 mode", 1e-4 against the fp64 oracle).  X and the sensitivity matrix are float, everything else double.

Two bars per result:
  * MODE  = 1e-4 against the oracle on the caller's fp64 X (the tolerance north_star states);
  * what the arithmetic actually promises against the oracle on X ROUNDED to float: the row operators carry
    every sum in fp64 (1e-12); the fused Hessian kernel multiplies in f32, takes its row dot in f32 and folds
    into fp64 every 4096 rows (1e-5); the TRSM chain is f32 throughout.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODE = 1e-4      # north_star's fp32-mode tolerance
F32 = 1e-5       # f32 MFMA products, chunked fp64 accumulation
EXACT = 1e-12    # fp64 sums of float data


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def _problem(N, D, family, seed, with_offset=False):
    from oracle import glm
    X, y, w, _ = glm.synth(N, D, family, seed)
    rng = np.random.default_rng(seed + 100)
    w = 0.5 + rng.random(N)
    theta = rng.standard_normal(D) / np.sqrt(D)
    offset = 0.1 * rng.standard_normal(N) if with_offset else None
    return X, y, w, theta, offset


@pytest.mark.parametrize("N,D", [(1, 1), (3, 5), (17, 16), (1000, 20), (257, 31), (4099, 32), (2000, 100),
                                 (513, 128), (1001, 97), (300, 113), (5003, 77), (20001, 100), (6001, 127),
                                 (40000, 81), (33000, 65)])
@pytest.mark.parametrize("family", ["logistic", "poisson", "gaussian"])
def test_fp32_fused_hessian_grad(N, D, family, relerr):
    torch = _torch()
    from oracle import glm
    import vittles_amd as va
    X, y, w, theta, offset = _problem(N, D, family, seed=N + D, with_offset=(D % 2 == 0))
    prior = np.linspace(0.5, 2.0, D)
    scale = 1.5
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior, scale=scale, offset=offset,
                                  dtype=torch.float32)
    assert obj.X.dtype == torch.float32
    out = obj.fused(theta, w)
    H, g = out["H"].cpu().numpy(), out["g"].cpu().numpy()
    assert out["H"].dtype == torch.float64 and np.array_equal(H, H.T)
    X32 = X.astype(np.float32).astype(np.float64)
    # against the oracle on the rounded X: the fused kernel's row dot is f32 (r and d are f32-accurate; the row
    # operator vt_glm_link_f32 gives them to fp64 accuracy, test_fp32_row_operators)
    _, l1, l2 = glm.link(family, glm.linear_predictor(X32, theta, offset), y)
    assert relerr(out["r"].cpu().numpy(), scale * l1) < F32
    assert relerr(out["d"].cpu().numpy(), scale * w * l2) < F32
    assert relerr(H, glm.hessian(family, X32, y, theta, w, prior, scale, offset)) < F32
    assert relerr(g, glm.grad(family, X32, y, theta, w, prior, scale, offset)) < F32
    # against the oracle on the caller's X: the mode's stated tolerance
    assert relerr(H, glm.hessian(family, X, y, theta, w, prior, scale, offset)) < MODE
    assert relerr(g, glm.grad(family, X, y, theta, w, prior, scale, offset)) < MODE


def test_fp32_hessian_empty_and_prior_only(relerr):
    torch = _torch()
    import vittles_amd as va
    D = 7
    prior = np.linspace(1.0, 2.0, D)
    theta = np.arange(D) / 10.0
    obj = va.WeightedGLMObjective(np.zeros((0, D)), np.zeros(0), prior=prior, dtype=torch.float32)
    out = obj.fused(theta, np.zeros(0))
    assert relerr(out["H"].cpu().numpy(), np.diag(prior)) == 0.0
    assert relerr(out["g"].cpu().numpy(), prior * theta) == 0.0


@pytest.mark.parametrize("D", [90, 20, 50])
def test_fp32_hessian_chunked_accumulation(D, relerr):
    """More than 4096 rows per wave: the f32 accumulators are folded into the fp64 slab several times
    (D = 90: one-wave-per-SIMD kernel, NB = 6, 1024 waves x 4096 rows = 4.2 M rows per fold; D = 20, 50: the
    row-link kernel, 2048 waves = 8.4 M rows per fold round).  Compared with the fp64 kernel on the same
    (float-representable) data; sizes beyond what the CPU oracle does in seconds."""
    torch = _torch()
    import vittles_amd as va
    N = 9_000_000
    gen = torch.Generator(device="cuda").manual_seed(5)
    X32 = torch.randn((N, D), generator=gen, device="cuda", dtype=torch.float32)
    theta = torch.randn(D, generator=gen, device="cuda", dtype=torch.float64) / D ** 0.5
    y = (torch.rand(N, generator=gen, device="cuda", dtype=torch.float64) < 0.5).to(torch.float64)
    w = 0.5 + torch.rand(N, generator=gen, device="cuda", dtype=torch.float64)
    o32 = va.WeightedGLMObjective(X32, y, prior=1.0, dtype=torch.float32).fused(theta, w)
    H32, g32 = o32["H"].cpu().numpy(), o32["g"].cpu().numpy()
    d32 = o32["d"][::1001].cpu().numpy()
    del o32
    o64 = va.WeightedGLMObjective(X32.to(torch.float64), y, prior=1.0).fused(theta, w)
    assert relerr(d32, o64["d"][::1001].cpu().numpy()) < F32
    assert relerr(H32, o64["H"].cpu().numpy()) < F32
    assert relerr(g32, o64["g"].cpu().numpy()) < MODE   # the gradient is a sum of signed terms near an optimum-free theta


@pytest.mark.parametrize("D", [20, 70])
def test_fp32_hessian_huge_leading_dimension(D, relerr):
    """X is a column block of a 4.2M-wide parent matrix: the row-link kernel's 32-bit descriptor offsets do not
    cover that stride and the dispatch must take the 64-bit-addressed kernel."""
    torch = _torch()
    import vittles_amd as va
    from vittles_amd import _lib
    lib = _lib.load()
    N, ld = 40, (1 << 22) + 8
    parent = torch.zeros((N, ld), dtype=torch.float32, device="cuda")
    gen = torch.Generator(device="cuda").manual_seed(9)
    parent[:, :D] = torch.randn((N, D), generator=gen, device="cuda", dtype=torch.float32)
    X = parent[:, :D]
    theta = torch.randn(D, generator=gen, device="cuda", dtype=torch.float64) / D ** 0.5
    y = (torch.rand(N, generator=gen, device="cuda", dtype=torch.float64) < 0.5).to(torch.float64)
    w = 0.5 + torch.rand(N, generator=gen, device="cuda", dtype=torch.float64)
    H = torch.empty((D, D), dtype=torch.float64, device="cuda")
    g = torch.empty(D, dtype=torch.float64, device="cuda")
    ws = torch.empty(lib.vt_glm_hessian_f32_workspace_bytes(N, D), dtype=torch.uint8, device="cuda")
    rc = lib.vt_glm_hessian_f32(0, X.data_ptr(), X.stride(0), theta.data_ptr(), y.data_ptr(), w.data_ptr(), None, 1.0,
                                None, N, D, H.data_ptr(), D, g.data_ptr(), None, None, ws.data_ptr(), ws.numel(),
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    ref = va.WeightedGLMObjective(X.contiguous().to(torch.float64), y, prior=0.0).fused(theta, w)
    assert relerr(H.cpu().numpy(), ref["H"].cpu().numpy()) < F32
    assert relerr(g.cpu().numpy(), ref["g"].cpu().numpy()) < MODE


def test_fp32_trsm_identity_layout():
    """H = I with an ASYMMETRIC integer right-hand side: the solve must return B element for element
    (catches any row / column swap in the f32 operand maps and in the pack conversion)."""
    torch = _torch()
    from vittles_amd.linalg import DeviceCholesky
    for D in (1, 16, 20, 100, 113, 128):
        chol = DeviceCholesky.factor(torch.eye(D, dtype=torch.float64, device="cuda"))
        B = (torch.arange(70 * D, dtype=torch.float32, device="cuda").reshape(70, D) + 1.0)
        out = chol.solve_rows(B)
        assert out.dtype == torch.float32
        assert torch.equal(out, B), D


@pytest.mark.parametrize("D", [1, 5, 16, 31, 48, 77, 100, 128])
def test_fp32_trsm_rows(D, relerr):
    torch = _torch()
    from vittles_amd.linalg import DeviceCholesky
    rng = np.random.default_rng(D)
    A = rng.standard_normal((D + 5, D))
    H = A.T @ A / D + np.eye(D)
    N = 1000 + D
    B = rng.standard_normal((N, D)).astype(np.float32)
    s = 0.5 + rng.random(N)
    chol = DeviceCholesky.factor(torch.tensor(H, device="cuda"))
    got = chol.solve_rows(torch.tensor(B, device="cuda"), rowscale=torch.tensor(s, device="cuda"), alpha=-2.0)
    ref = -2.0 * s[:, None] * np.linalg.solve(H, B.astype(np.float64).T).T
    assert got.dtype == torch.float32 and got.shape == (N, D)
    assert relerr(got.cpu().numpy(), ref) < F32 * 10


@pytest.mark.parametrize("N,D", [(1, 3), (16, 16), (33, 20), (1000, 20), (4099, 100), (777, 128), (500, 45)])
@pytest.mark.parametrize("family", ["logistic", "gaussian"])
def test_fp32_sensitivity_end_to_end(N, D, family, relerr):
    """The drop-in class in fp32 mode against the fp64 oracle (reference sensitivity_lib.py:226), 1e-4."""
    torch = _torch()
    from oracle import glm
    import vittles_amd as va
    X, y, w, _, _ = _problem(N, D, family, seed=7 * N + D)
    prior = 1.0
    theta = glm.newton_optimum(family, X, y, w, prior=prior)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior, dtype=torch.float32)
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
    H, jac0, S = glm.sensitivity(family, X, y, theta, w, prior=prior)
    got = sens.get_dopt_dhyper()
    assert got.shape == (D, N) and got.dtype == np.float32
    assert relerr(got, S) < MODE
    assert relerr(sens.get_hessian_at_opt(), H) < MODE
    w1 = w * (1.0 + 0.01 * np.cos(np.arange(N)))
    pred = sens.predict_opt_par_from_hyper_par(w1)
    assert pred.dtype == np.float64
    assert relerr(pred, theta + S @ (w1 - w)) < MODE
    # validate_optimum: the gradient of the ROUNDED problem at the fp64 optimum is ~1e-7 N, not 1e-8
    va.HyperparameterSensitivityLinearApproximation(obj, theta, w, validate_optimum=True, grad_tol=1e-3 * N)


def test_fp32_row_operators(relerr):
    """K0 link, K3 gradient, K6 HVP, K9 A^T u, A v, value on float X: fp64 sums of float data, so they agree
    with the oracle on the rounded X to fp64 accuracy."""
    torch = _torch()
    from oracle import glm
    from vittles_amd import rows
    import vittles_amd as va
    for (N, D, family) in [(1000, 20, "logistic"), (513, 129, "poisson"), (3000, 1000, "gaussian"),
                           (257, 31, "logistic"), (64, 4096, "logistic"), (5000, 100, "poisson"),
                           (300, 3001, "logistic"),     # odd D in (2048, 4096]: rows re-laid on a 16-byte pitch
                           # wide float rows (round 3: the wide team on 16-byte lanes): D > 4096, odd widths re-laid
                           (300, 4100, "logistic"), (257, 10241, "gaussian"), (200, 20000, "logistic"),
                           (130, 20447, "poisson")]:
        X, y, w, theta, offset = _problem(N, D, family, seed=N + D, with_offset=True)
        if family == "poisson":
            theta = theta * (0.1 if D < 4096 else 0.01)
        X32 = X.astype(np.float32).astype(np.float64)
        prior = np.linspace(0.5, 2.0, D)
        obj = va.WeightedGLMObjective(X, y, family=family, prior=prior, scale=0.7, offset=offset,
                                      dtype=torch.float32)
        th, wd = torch.tensor(theta, device="cuda"), torch.tensor(w, device="cuda")
        z, r, d = rows.glm_link(obj, th, wd, want_z=True)
        zr = glm.linear_predictor(X32, theta, offset)
        _, l1, l2 = glm.link(family, zr, y)
        assert relerr(z.cpu().numpy(), zr) < EXACT
        assert relerr(r.cpu().numpy(), 0.7 * l1) < EXACT
        g, _, d2 = rows.glm_grad(obj, th, wd, prior=obj.prior)
        assert relerr(g.cpu().numpy(), glm.grad(family, X32, y, theta, w, prior, 0.7, offset)) < 1e-11
        v = np.cos(np.arange(D))
        hv = rows.glm_hvp(obj.X, d2, torch.tensor(v, device="cuda"), prior=obj.prior)
        assert relerr(hv.cpu().numpy(), glm.hvp(family, X32, y, theta, w, v, prior, 0.7, offset)) < 1e-11
        u = np.sin(np.arange(N))
        assert relerr(rows.gemv_t(obj.X, torch.tensor(u, device="cuda")).cpu().numpy(), X32.T @ u) < 1e-11
        assert relerr(rows.gemv(obj.X, torch.tensor(v, device="cuda")).cpu().numpy(), X32 @ v) < EXACT
        u2 = np.cos(0.3 * np.arange(N))
        assert relerr(rows.gemv_t2(obj.X, torch.tensor(u, device="cuda"), torch.tensor(u2, device="cuda")).cpu().numpy(),
                      X32.T @ (u * u2)) < 1e-11
        val = obj(theta, w)
        ref = glm.value(family, X32, y, theta, w, prior, 0.7, offset)
        assert abs(val - ref) <= 1e-11 * abs(ref)


def test_fp32_cg_config4_d20000(relerr):
    """BASELINE.json configs[3] (matrix-free CG, D = 20 000 ridge-logistic) on a FLOAT X: the wide row kernel on
    16-byte lanes.  fp64 sums of float data: the device iteration tracks the oracle's CG on the rounded X step for
    step, and lands within the mode's 1e-4 of the fp64 problem's solution."""
    torch = _torch()
    import vittles_amd as va
    from vittles_amd.cg import GLMHessianVectorProduct
    from vittles_amd import rows
    from oracle import glm, solvers
    N, D = 3000, 20000
    X, y, w, _ = glm.synth(N, D, "logistic", seed=20)
    X32 = X.astype(np.float32).astype(np.float64)
    rng = np.random.default_rng(2)
    theta = rng.standard_normal(D) / np.sqrt(D)
    w = rng.uniform(0.5, 1.5, N)
    dw = rng.standard_normal(N)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0, dtype=torch.float32)
    assert obj.X.dtype == torch.float32
    hvp = GLMHessianVectorProduct(obj, theta, w)
    _, r, _ = rows.glm_link(obj, torch.tensor(theta, device="cuda"), torch.tensor(w, device="cuda"))
    b = rows.gemv_t2(obj.X, r, torch.tensor(dw, device="cuda")).cpu().numpy()       # jac0 @ dw
    sols = {}
    for name, Xr in (("rounded", X32), ("fp64", X)):
        _, l1, l2 = glm.link("logistic", Xr @ theta, y)
        mv = lambda u, Xr=Xr, dv=w * l2: Xr.T @ (dv * (Xr @ u)) + u
        br = Xr.T @ (l1 * dw)              # d grad / d w_n = ell_n' x_n: no weight factor
        sols[name] = (br, solvers.cg_legacy(mv, br, tol=1e-10, return_iters=True))
    assert relerr(b, sols["rounded"][0]) < 1e-11
    x = va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': 1e-10})(b)
    x_ref, info_ref, _ = sols["rounded"][1]
    assert info_ref == 0 and relerr(x, x_ref) < 1e-8
    assert relerr(x, sols["fp64"][1][0]) < MODE


def test_fp32_newton_fit(relerr):
    """The optimiser runs on the float X (gradient / Hessian / value through the *_f32 entry points) and lands
    on the optimum of the rounded problem."""
    torch = _torch()
    from oracle import glm
    import vittles_amd as va
    from vittles_amd.optim import newton_glm
    N, D = 6000, 40
    X, y, w, _, _ = _problem(N, D, "logistic", seed=11)
    obj = va.WeightedGLMObjective(X, y, prior=1.0, dtype=torch.float32)
    res = newton_glm(obj, w)
    theta = res[0] if isinstance(res, tuple) else res
    theta = theta.cpu().numpy() if hasattr(theta, "cpu") else np.asarray(theta)
    X32 = X.astype(np.float32).astype(np.float64)
    assert relerr(theta, glm.newton_optimum("logistic", X32, y, w, prior=1.0)) < 1e-6
    assert relerr(theta, glm.newton_optimum("logistic", X, y, w, prior=1.0)) < MODE


def test_fp32_limits_raise_clearly():
    torch = _torch()
    import vittles_amd as va
    from vittles_amd._lib import VittlesHipError
    X, y, w, theta, _ = _problem(30, 4100, "logistic", seed=2)      # past the float row kernels' D <= 4096
    obj = va.WeightedGLMObjective(X, y, prior=1.0, dtype=torch.float32)
    with pytest.raises(VittlesHipError) as e:
        obj.fused(theta, w)
    assert "fp32 mode" in str(e.value)
    # the Taylor expansion is fp64-only
    X, y, w, theta, _ = _problem(300, 10, "logistic", seed=2)
    obj = va.WeightedGLMObjective(X, y, prior=1.0, dtype=torch.float32)
    with pytest.raises(VittlesHipError) as e:
        va.ParametricSensitivityTaylorExpansion.optimization_objective(obj, theta, w, order=2).evaluate_taylor_series(w)
    assert "fp32 storage mode" in str(e.value)
    with pytest.raises(ValueError):
        va.WeightedGLMObjective(X, y, dtype=torch.float16)


@pytest.mark.parametrize("N,Da,Db", [(5000, 3, 20), (70000, 130, 100), (999, 64, 257), (4097, 17, 33), (1, 1, 1)])
def test_fp32_gemm_tn_float_operand(N, Da, Db):
    """W^T diag(s) S + bias with S stored in float: the products and sums are fp64, so the result equals the fp64
    kernel's on the same (float-representable) values bit for bit; odd Db exercises the path without LDS-DMA."""
    torch = _torch()
    from vittles_amd import rows
    gen = torch.Generator(device="cuda").manual_seed(N + Da)
    A = torch.randn((N, Da), generator=gen, device="cuda", dtype=torch.float64)
    B32 = torch.randn((N, Db), generator=gen, device="cuda", dtype=torch.float32)
    s = 0.5 + torch.rand(N, generator=gen, device="cuda", dtype=torch.float64)
    bias = torch.randn(Db, generator=gen, device="cuda", dtype=torch.float64)
    got = rows.gemm_tn(A, B32, rowscale=s, alpha=-1.5, bias=bias)
    want = rows.gemm_tn(A, B32.to(torch.float64), rowscale=s, alpha=-1.5, bias=bias)
    assert got.dtype == torch.float64
    assert torch.equal(got, want)


def test_fp32_predict_batch_matches_loop(relerr):
    """fp32 storage mode: the batched prediction (one W^T S product over the float S) equals the one-vector
    prediction column by column, and the fp64 pipeline's to the mode's tolerance."""
    torch = _torch()
    import vittles_amd as va
    X, y, w, theta, _ = _problem(6000, 40, "logistic", 21)
    Xd = torch.tensor(X, device="cuda")
    yd, wd = torch.tensor(y, device="cuda"), torch.tensor(w, device="cuda")
    obj64 = va.WeightedGLMObjective(Xd, yd, prior=1.0)
    theta_hat, _ = va.optim.newton_glm(obj64, wd, opt_par_init=torch.tensor(theta, device="cuda"))
    obj32 = va.WeightedGLMObjective(Xd, yd, prior=1.0, dtype=torch.float32)
    lin32 = va.HyperparameterSensitivityLinearApproximation(obj32, theta_hat, wd)
    lin64 = va.HyperparameterSensitivityLinearApproximation(obj64, theta_hat, wd)
    rng = np.random.default_rng(3)
    W = torch.tensor(w[:, None] * rng.exponential(size=(6000, 7)), device="cuda")
    got = lin32.predict_opt_par_from_hyper_par_batch(W)
    assert tuple(got.shape) == (40, 7)
    for k in range(7):
        one = lin32.predict_opt_par_from_hyper_par(W[:, k])
        assert relerr(got[:, k].cpu().numpy(), one.cpu().numpy()) < EXACT
    assert relerr(got.cpu().numpy(), lin64.predict_opt_par_from_hyper_par_batch(W).cpu().numpy()) < MODE


# ---- D > 128 in the fp32 storage mode: float X and S, tiled SYRK on float panels with fp64
# products, slab-wise fp64 solve -- configs[2]'s D = 1000 at half the footprint ----------------------------------
@pytest.mark.parametrize("N,D,family", [(3000, 129, "logistic"), (5000, 160, "poisson"), (4100, 257, "gaussian"),
                                         (6001, 1000, "logistic"), (2500, 1001, "logistic"), (4000, 384, "logistic")])
def test_fp32_large_d_hessian_and_sensitivity(N, D, family, relerr):
    torch = _torch()
    from oracle import glm
    import vittles_amd as va
    X, y, w, _, _ = _problem(N, D, family, seed=N + D)
    prior = 1.0
    theta = glm.newton_optimum(family, X, y, w, prior=prior)
    obj = va.WeightedGLMObjective(X, y, family=family, prior=prior, dtype=torch.float32)
    assert obj.X.dtype == torch.float32 and obj.X.stride(0) % 4 == 0
    out = obj.fused(theta, w)
    H = out["H"].cpu().numpy()
    assert out["H"].dtype == torch.float64 and np.array_equal(H, H.T)
    # the products and sums are fp64: against the oracle on the ROUNDED X the Hessian is fp64-exact
    X32 = X.astype(np.float32).astype(np.float64)
    assert relerr(H, glm.hessian(family, X32, y, theta, w, prior)) < 1e-10
    assert relerr(out["g"].cpu().numpy() + 1.0, glm.grad(family, X32, y, theta, w, prior) + 1.0) < 1e-10
    # and within the mode's tolerance of the fp64 oracle on the caller's X
    H_ref, _, S_ref = glm.sensitivity(family, X, y, theta, w, prior=prior)
    assert relerr(H, H_ref) < MODE
    sens = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
    S = sens.get_dopt_dhyper()
    assert S.shape == (D, N) and sens._sens_rows.dtype == torch.float32
    assert relerr(S, S_ref) < MODE
    w1 = w * (1.0 + 0.01 * np.cos(np.arange(N)))
    assert relerr(sens.predict_opt_par_from_hyper_par(w1), theta + S_ref @ (w1 - w)) < MODE


def test_fp32_large_d_solve_in_slabs(relerr):
    """vt_trsm_rows_f32_large with a scratch smaller than the matrix: several slabs, ragged last one."""
    torch = _torch()
    import scipy.linalg
    from vittles_amd import _lib, device as dev
    from vittles_amd.linalg import DeviceCholesky
    rng = np.random.default_rng(5)
    D, N = 300, 20000
    A = rng.standard_normal((2 * D, D))
    Hm = A.T @ A / D + np.eye(D)
    chol = DeviceCholesky.factor(torch.tensor(Hm, device="cuda"), pack=True)
    B = rng.standard_normal((N, D)).astype(np.float32)
    s = 0.5 + rng.random(N)
    Bd, sd = torch.tensor(B, device="cuda"), torch.tensor(s, device="cuda")
    out = torch.empty((N, D), dtype=torch.float32, device="cuda")
    ws = torch.empty(6000 * D, dtype=torch.float64, device="cuda")        # 6000 rows per slab: 4 slabs
    lib = _lib.load()
    _lib.check(lib.vt_trsm_rows_f32_large(chol.pack().data_ptr(), D, Bd.data_ptr(), D, sd.data_ptr(), -2.0, N,
                                          out.data_ptr(), D, ws.data_ptr(), ws.numel() * 8, dev.stream_ptr()), "slabs")
    ref = -2.0 * s[:, None] * scipy.linalg.cho_solve(scipy.linalg.cho_factor(Hm), B.astype(np.float64).T).T
    assert relerr(out.cpu().numpy().astype(np.float64), ref) < 1e-6
    assert torch.equal(chol.solve_rows(Bd, rowscale=sd, alpha=-2.0), out)   # the one-slab route gives the same floats
    tiny = torch.empty(10 * D, dtype=torch.float64, device="cuda")
    assert lib.vt_trsm_rows_f32_large(chol.pack().data_ptr(), D, Bd.data_ptr(), D, 0, 1.0, N, out.data_ptr(), D,
                                      tiny.data_ptr(), tiny.numel() * 8, dev.stream_ptr()) == -3   # VT_ERR_WORKSPACE
