"""GPU parity of the block-sparse VI path (K7/K8): registered hierarchical mean-field Gaussian
objective, SparseBlockHessian assembly (COO entries in the reference's order) and
LinearResponseCovariances on the block-arrowhead Hessian, against the oracle."""
import numpy as np
import pytest
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _model(G, ng, k, seed=0, taus=(2.0, 1.5, 0.5)):
    import vittles_amd as va
    from oracle import vi
    X, y = vi.synth(G, ng, k, seed)
    m = vi.HierGaussMFVI(X, y, G, ng, k, *taus)
    obj = va.HierGaussMFVIObjective(X, y, G, ng, k, *taus)
    return m, obj


@pytest.mark.parametrize("G,ng,k", [(1, 1, 1), (6, 5, 2), (37, 9, 3), (300, 16, 4)])
def test_vi_objective_derivatives(G, ng, k, relerr):
    m, obj = _model(G, ng, k, seed=G)
    rng = np.random.default_rng(G)
    eta = m.optimum() * (1.0 + 0.05 * rng.random(m.d))
    w = 0.5 + rng.random(G * ng)
    u = rng.standard_normal(m.d)
    assert abs(obj(eta, w) - m.value(eta, w)) <= TOL * abs(m.value(eta, w))
    assert relerr(obj.grad(eta, w).cpu().numpy(), m.grad(eta, w)) < TOL
    assert relerr(obj.hvp(eta, u, w).cpu().numpy(), m.hvp(eta, u, w)) < TOL
    assert abs(obj(eta) - m.value(eta)) <= TOL * abs(m.value(eta))


@pytest.mark.parametrize("G,ng,k", [(6, 5, 2), (50, 8, 3)])
def test_sparse_block_hessian_matches_reference_order(G, ng, k, relerr):
    import vittles_amd as va
    from oracle import block_hessian
    m, obj = _model(G, ng, k, seed=3)
    eta = m.optimum() * 1.05
    ref = block_hessian.SparseBlockHessian(lambda o, v: m.hvp(o, v), m.sparsity_array())
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())
    blk, blk_ref = sb.get_block_hessian(eta), ref.get_block_hessian(eta)
    assert isinstance(blk, scipy.sparse.coo_matrix) and blk.shape == (m.d, m.d)
    assert np.array_equal(blk.row, blk_ref.row) and np.array_equal(blk.col, blk_ref.col)   # same entry ORDER
    assert relerr(blk.data, blk_ref.data) < TOL
    glb, glb_ref = sb.get_global_hessian(eta), ref.get_global_hessian(eta)
    assert np.array_equal(glb.row, glb_ref.row) and np.array_equal(glb.col, glb_ref.col)
    assert relerr(glb.data, glb_ref.data) < TOL
    glb2 = sb.get_global_hessian(eta, global_inds=obj.global_inds())
    assert relerr(glb2.data, glb_ref.data) < TOL
    Hd = m.hessian(eta)
    # tests/test_sparse_hessian_lib.py:101-113
    assert relerr(np.asarray((blk + glb).todense()), Hd) < TOL
    assert relerr(np.asarray(sb.get_hessian(eta, print_every=0).todense()), Hd) < TOL
    with pytest.raises(ValueError, match="must be unique"):
        va.SparseBlockHessian(obj, np.array([[0, 1], [1, 2]]))
    with pytest.raises(ValueError, match="must be disjoint"):
        sb.get_global_hessian(eta, global_inds=np.array([0]))
    with pytest.raises(ValueError, match="must be a vector"):
        sb.get_block_hessian(eta[None, :])


def test_block_only_objective_separable(relerr):
    """tests/test_sparse_hessian_lib.py:15-53: no globals -> block Hessian == full Hessian."""
    import torch
    import vittles_amd as va

    class Separable:
        """f(x) = 1/2 sum_g x_g^T A_g x_g  with fixed SPD A_g: hvp is blockwise A_g v_g."""
        def __init__(self, A):
            self.A = torch.tensor(A, device="cuda")
            self.device = self.A.device
        def hvp(self, opt, v):
            G, M, _ = self.A.shape
            return torch.einsum('gij,gj->gi', self.A, v.reshape(G, M)).reshape(-1)
    rng = np.random.default_rng(42)
    G, M = 10, 9
    R = rng.standard_normal((G, M, M))
    A = R @ R.transpose(0, 2, 1) + M * np.eye(M)
    perm = rng.permutation(G * M).reshape(G, M)        # blocks "up to a permutation" (reference :17-24)
    Hd = np.zeros((G * M, G * M))
    for g in range(G):
        Hd[np.ix_(perm[g], perm[g])] = A[g]

    class Permuted(Separable):
        def hvp(self, opt, v):
            vv = v[torch.tensor(perm.reshape(-1), device="cuda")]
            out = torch.zeros_like(v)
            out[torch.tensor(perm.reshape(-1), device="cuda")] = super().hvp(opt, vv)
            return out
    sb = va.SparseBlockHessian(Permuted(A), perm)
    x = rng.standard_normal(G * M)
    h = sb.get_block_hessian(x)
    assert relerr(np.asarray(h.todense()), Hd) < TOL
    # and the structured solver on it
    b = rng.standard_normal((G * M, 3))
    sol = va.solver_lib.get_cholesky_solver(h)(b)
    assert relerr(sol, np.linalg.solve(Hd, b)) < 1e-9


@pytest.mark.parametrize("G,ng,k", [(4, 6, 2), (40, 8, 3)])
def test_lr_covariance_block_arrowhead(G, ng, k, relerr):
    import vittles_amd as va
    from vittles_amd.lr_cov_lib import LinearMoments
    from oracle import lr_cov
    m, obj = _model(G, ng, k, seed=1, taus=(1.3, 0.7, 0.4))
    eta = m.optimum()
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())
    H_sparse = sb.get_hessian(eta)
    lr = va.LinearResponseCovariances(obj, eta, validate_optimum=True, hessian_at_opt=H_sparse, grad_tol=1e-8)
    assert lr.get_hessian_at_opt() is H_sparse
    mean_idx = np.concatenate([np.arange(g * 2 * k, g * 2 * k + k) for g in range(G)] +
                              [np.arange(2 * k * G, 2 * k * G + k)])
    J = np.zeros((len(mean_idx), m.d))
    J[np.arange(len(mean_idx)), mean_idx] = 1.0
    cov = lr.get_lr_covariance_from_jacobians(J, J)
    # oracle: same formula with the reference's SuperLU closure on the same matrix
    ref = lr_cov.LinearResponseCovariances(m.grad, m.hessian, eta, hessian_at_opt=scipy.sparse.csc_matrix(m.hessian(eta)))
    assert relerr(cov, ref.get_lr_covariance_from_jacobians(J, J)) < 1e-9
    # exactness for a Gaussian model (tests/test_lr_cov_lib.py:90-93): LR covariance == posterior covariance
    P = np.zeros((k * (G + 1), k * (G + 1)))
    Xg = m.X
    for g in range(G):
        P[g * k:(g + 1) * k, g * k:(g + 1) * k] = 1.3 * Xg[g].T @ Xg[g] + 0.7 * np.eye(k)
        P[g * k:(g + 1) * k, G * k:] = -0.7 * np.eye(k)
        P[G * k:, g * k:(g + 1) * k] = -0.7 * np.eye(k)
    P[G * k:, G * k:] = (G * 0.7 + 0.4) * np.eye(k)
    assert relerr(cov, np.linalg.inv(P)) < 1e-9
    assert relerr(lr.get_lr_covariance(LinearMoments(J)), cov) < 1e-12
    assert relerr(lr.get_lr_covariance_from_jacobians(J[:3], J[5:9]), cov[:3, 5:9]) < 1e-9
    # shape errors (tests/test_lr_cov_lib.py:118-135)
    for bad in ((J.T, J), (J, J.T), (J[:, :, None], J), (J, J[:, :, None])):
        with pytest.raises(ValueError):
            lr.get_lr_covariance_from_jacobians(*bad)
    with pytest.raises(ValueError, match="gradient is not zero"):
        va.LinearResponseCovariances(obj, eta * 1.01, validate_optimum=True, hessian_at_opt=H_sparse)


def test_lr_covariance_reference_kat_dense():
    """tests/test_lr_cov_lib.py:16-98 (free=False): dense H, LR covariance of the means == Sigma."""
    import torch
    import vittles_amd as va
    dim = 4
    mu = np.arange(0, dim).astype(float)
    cov = dim * np.eye(dim) + np.outer(mu, mu)
    info = np.linalg.inv(cov)
    var = 1 / np.diag(info)
    opt = np.concatenate([mu, var])
    H = scipy.linalg.block_diag(info, np.diag(0.5 / var ** 2))

    class KL:
        def grad(self, eta):
            e = eta.cpu().numpy()
            g = np.concatenate([info @ (e[:dim] - mu), 0.5 * np.diag(info) - 0.5 / e[dim:]])
            return torch.tensor(g, device="cuda")
        def hessian(self, eta):
            return torch.tensor(H, device="cuda")
    for init_h in (False, True):
        lr = va.LinearResponseCovariances(KL(), opt, validate_optimum=True,
                                          hessian_at_opt=H if init_h else None, grad_tol=1e-13)
        np.testing.assert_array_almost_equal(H, lr.get_hessian_at_opt())
        J = np.hstack([np.eye(dim), np.zeros((dim, dim))])
        c = lr.get_lr_covariance_from_jacobians(J, J)
        np.testing.assert_array_almost_equal(cov, c)
        np.testing.assert_array_almost_equal(c[0:2, 2:4], lr.get_lr_covariance_from_jacobians(J[0:2], J[2:4]))


def test_config5_scale_consistency():
    """BASELINE.json configs[4] size: G = 10^4 groups.  Size-independent properties: the
    structured solve satisfies H x = b (checked through HVPs), and the gradient vanishes at the
    closed-form optimum."""
    import time
    import torch
    import vittles_amd as va
    G, ng, k = 10_000, 16, 4
    m, obj = _model(G, ng, k, seed=5)
    eta = m.optimum()
    assert float(obj.grad(eta).norm()) < 1e-8
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())
    t0 = time.perf_counter()
    H = sb.get_hessian(eta)
    t1 = time.perf_counter()
    assert H.shape == (obj.d, obj.d)
    solve = va.solver_lib.get_cholesky_solver(H)
    b = torch.randn(obj.d, 3, dtype=torch.float64, device="cuda")
    x = solve(b)
    for c in range(3):
        resid = obj.hvp(eta, x[:, c].contiguous()) - b[:, c]
        assert float(resid.norm() / b[:, c].norm()) < 1e-10
    print("config5: d=%d assemble %.3fs" % (obj.d, t1 - t0))


def test_sparse_solver_recovers_structure_from_the_matrix():
    """reference solver_lib.py:33-48 takes ANY sparse matrix.  At BASELINE.json configs[4] size (d = 80 008) every
    route by which the private structure attribute gets lost -- .tocsc(), .copy(), H1 + H2, a COO assembled by the
    user, an index permutation -- still takes the batched block-arrowhead kernels (structure recovered from the
    sparsity pattern) and gives the attribute path's solution; the residual is checked through HVPs."""
    import torch
    import vittles_amd as va
    from vittles_amd import sparse_hessian_lib as shl
    G, ng, k = 10_000, 16, 4
    m, obj = _model(G, ng, k, seed=5)
    eta = m.optimum()
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())
    H = sb.get_hessian(eta)
    assert getattr(H, "_vt_structure", None) is not None
    b = np.random.default_rng(0).standard_normal((obj.d, 2))
    x_attr = va.solver_lib.get_cholesky_solver(H)(b)

    def check(mat, label):
        assert getattr(mat, "_vt_structure", None) is None, label        # the attribute really is gone
        x = va.solver_lib.get_cholesky_solver(mat)(b)
        assert np.abs(x - x_attr).max() <= 1e-12 * np.abs(x_attr).max(), label
        for c in range(2):
            resid = obj.hvp(eta, x[:, c]).cpu().numpy() - b[:, c]
            assert np.linalg.norm(resid) < 1e-10 * np.linalg.norm(b[:, c]), label

    check(H.tocsc(), "tocsc")
    check(scipy.sparse.csr_matrix(H).copy(), "csr copy")
    check(sb.get_block_hessian(eta) + sb.get_global_hessian(eta), "H1 + H2")
    coo = H.tocoo()
    check(scipy.sparse.coo_matrix((coo.data.copy(), (coo.row.copy(), coo.col.copy())), H.shape), "user COO")
    # explicit keywords (build-side extras) give the same solver
    x_kw = va.solver_lib.get_sparse_cholesky_solver(H.tocsc(), sparsity_array=obj.sparsity_array(),
                                                    global_inds=obj.global_inds())(b)
    assert np.abs(x_kw - x_attr).max() <= 1e-12 * np.abs(x_attr).max()
    # blocks that are NOT contiguous index ranges: a symmetric permutation of the matrix
    perm = np.random.default_rng(1).permutation(obj.d)
    P = scipy.sparse.csr_matrix((np.ones(obj.d), (np.arange(obj.d), perm)), (obj.d, obj.d))
    Hp = (P @ H.tocsr() @ P.T).tocsc()
    found = shl.detect_arrowhead(Hp)
    # with explicit zeros gone the pattern shows the k coupled means of a group as one component, every variance
    # (and the globals' variances) as singletons, and only the k global means as dense rows: singletons are
    # packed k to a block, no padding needed
    assert found is not None and found[0].shape == (2 * G + 1, k) and len(found[1]) == k and found[2] == obj.d
    xp = va.solver_lib.get_cholesky_solver(Hp)(P @ b)
    assert np.abs(P.T @ xp - x_attr).max() <= 1e-11 * np.abs(x_attr).max()


def test_sparse_solver_stale_snapshot_and_fallbacks():
    """(b) an in-place edit of the scipy matrix must not be answered from the stale device snapshot;
    (a) blocks wider than the batched kernels take fall back to the dense route when the matrix is small, and
    anything unstructured above the dense limit raises instead of densifying tens of GB."""
    import vittles_amd as va
    from vittles_amd._lib import VittlesHipError
    m, obj = _model(400, 6, 2, seed=9)
    eta = m.optimum()
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())
    H = sb.get_hessian(eta)
    b = np.random.default_rng(3).standard_normal(obj.d)
    x0 = va.solver_lib.get_cholesky_solver(H)(b)
    H.data *= 2.0                                           # attribute survives, values changed
    x1 = va.solver_lib.get_cholesky_solver(H)(b)
    assert np.abs(x1 - 0.5 * x0).max() <= 1e-12 * np.abs(x0).max()
    Hd = np.asarray(H.todense())
    assert np.abs(x1 - np.linalg.solve(Hd, b)).max() <= 1e-10 * np.abs(x1).max()
    # (a) one block of 100 > 64: small matrix -> dense route, same answer as SciPy
    rng = np.random.default_rng(4)
    A = rng.standard_normal((300, 300))
    A = scipy.sparse.csc_matrix(np.kron(np.eye(3), np.ones((100, 100))) * (A @ A.T) + 300 * np.eye(300))
    v = rng.standard_normal(300)
    xs = va.solver_lib.get_sparse_cholesky_solver(A)(v)
    assert np.abs(xs - scipy.sparse.linalg.spsolve(A, v)).max() <= 1e-10 * np.abs(xs).max()
    # unstructured (a chain: one connected component) above the dense limit: the iterative route (block CG over
    # vt_spmv_csr, Jacobi preconditioner), never a todense(); the reference hands this to SuperLU (solver_lib.py:46-48)
    n = 9000
    T = scipy.sparse.diags([np.full(n - 1, -1.0), np.full(n, 4.0), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsc()
    solveT = va.solver_lib.get_sparse_cholesky_solver(T)
    vT = rng.standard_normal(n)
    xT = solveT(vT)
    assert np.abs(xT - scipy.sparse.linalg.spsolve(T, vT)).max() <= 1e-10 * np.abs(xT).max()
    VT = rng.standard_normal((n, 5))
    XT = solveT(VT)
    assert XT.shape == (n, 5) and np.abs(T @ XT - VT).max() <= 1e-10 * np.abs(VT).max()
    # a random unstructured SPD pattern with uneven rows and a badly scaled diagonal
    Rm = scipy.sparse.random(n, n, density=4e-4, random_state=7, format="csr")
    Rm = Rm + Rm.T
    Dg = scipy.sparse.diags(np.logspace(0, 3, n))
    Am = (Dg @ (Rm + scipy.sparse.diags(np.asarray(abs(Rm).sum(axis=1)).ravel() + 1.0)) @ Dg).tocsc()
    xa = va.solver_lib.get_sparse_cholesky_solver(Am)(vT)
    assert np.abs(xa - scipy.sparse.linalg.spsolve(Am, vT)).max() <= 1e-9 * np.abs(xa).max()
    # not symmetric -> ValueError; indefinite -> LinAlgError (no silent poor answer)
    with pytest.raises(ValueError, match="SYMMETRIC"):
        va.solver_lib.get_sparse_cholesky_solver(scipy.sparse.triu(T).tocsc())
    Tind = (T - scipy.sparse.diags(np.where(np.arange(n) % 2 == 0, 6.0, 0.0))).tocsc()
    with pytest.raises(np.linalg.LinAlgError):
        va.solver_lib.get_sparse_cholesky_solver(Tind)(vT)
    Tind2 = (T - 3.9 * scipy.sparse.identity(n)).tocsc()       # positive diagonal, eigenvalues of both signs
    with pytest.raises(np.linalg.LinAlgError):
        va.solver_lib.get_sparse_cholesky_solver(Tind2)(vT)
    # ... and a non-SPD block matrix raises LinAlgError (Cholesky only; the reference's SuperLU would take it)
    Hneg = H.copy()
    Hneg.data = -Hneg.data
    with pytest.raises(np.linalg.LinAlgError):
        va.solver_lib.get_cholesky_solver(Hneg)(b)


def test_config5_pipeline_launches_no_library_gemm(relerr):
    """the config-5 pipeline (block Hessian -> LR covariances, also with more moments than one
    vt_gemm_tn call takes) and the dense sensitivity path run under the kernel profiler: every device kernel is one
    of this library's (k_*) or a torch copy / fill -- no rocBLAS / Tensile GEMM, no torch reduction kernel."""
    import torch
    import vittles_amd as va
    from torch.profiler import profile, ProfilerActivity
    G, ng, k = 40, 6, 3
    m, obj = _model(G, ng, k, seed=5)
    eta = m.optimum()
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())
    rng = np.random.default_rng(0)
    J = rng.standard_normal((12, m.d))
    from oracle import glm
    X, y, w, _ = glm.synth(3000, 40, "logistic", seed=2)
    theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)

    def pipeline():
        H = sb.get_hessian(eta)
        lr = va.LinearResponseCovariances(obj, eta, validate_optimum=True, hessian_at_opt=H, grad_tol=1e-6)
        cov = lr.get_lr_covariance_from_jacobians(J, J)
        gobj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
        sens = va.HyperparameterSensitivityLinearApproximation(gobj, torch.tensor(theta, device="cuda"),
                                                               torch.tensor(w, device="cuda"), validate_optimum=True)
        sens.predict_opt_par_from_hyper_par(torch.tensor(w * 1.01, device="cuda"))
        return cov
    pipeline()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        cov = pipeline()
        torch.cuda.synchronize()
    names = sorted({e.name for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")})
    if not names:
        pytest.skip("the kernel profiler returned no device events on this box")
    ours = lambda n: "(anonymous namespace)::k_" in n             # this library's kernels (csrc/*.hip)
    bad = [n for n in names if not ours(n) and any(t in n.lower() for t in ("gemm", "gemv", "rocblas", "cijk", "tensile",
                                                                             "reduce_kernel", "dot_kernel"))]
    assert not bad, bad
    assert sum(ours(n) for n in names) >= 8, names
    Hc = np.asarray(sb.get_hessian(eta).todense())
    assert relerr(cov, J @ np.linalg.solve(Hc, J.T)) < 1e-9
    # more moments than one vt_gemm_tn call takes: column panels of 8192, still no library GEMM
    from vittles_amd import rows
    A = torch.tensor(rng.standard_normal((300, 8200)), device="cuda")
    B = torch.tensor(rng.standard_normal((300, 5)), device="cuda")
    assert relerr(rows.gemm_tn_tiled(A, B).cpu().numpy(), A.cpu().numpy().T @ B.cpu().numpy()) < 1e-10
