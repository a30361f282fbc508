"""One rank of the 2-rank product-path test (started by tests/test_gpu_multirank.py as a fresh
process with RANK / WORLD_SIZE / MASTER_* set).  Both ranks share GPU 0 and talk over gloo
(RCCL refuses two ranks on one device); device tensors are staged through the host by
vittles_amd.sharding, everything else is the code path that runs over RCCL on an 8-GPU node.

Checks the ROW-SHARDED product classes against the oracle on the unsharded data."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300))


def glm_checks(rank, world, group, family, N, D):
    import torch
    import vittles_amd as va
    from vittles_amd import sharding
    from vittles_amd.cg import GLMHessianVectorProduct, cg_solve
    from oracle import glm, solvers, taylor
    X, y, w, _ = glm.synth(N, D, family, seed=100 + D)
    rng = np.random.default_rng(5)
    w = rng.uniform(0.5, 1.5, N)
    prior = 1.5
    a, b = sharding.shard_rows(N, rank, world)
    obj = va.WeightedGLMObjective(X[a:b], y[a:b], family=family, prior=prior, process_group=group)

    # optimiser: every rank takes identical steps
    theta_ref = glm.newton_optimum(family, X, y, w, prior=prior)
    theta, info = va.optim.newton_glm(obj, w[a:b], grad_tol=1e-10)
    assert info["converged"], info
    assert rel(theta, theta_ref) < 1e-10, ("newton", rel(theta, theta_ref))

    # fused pass: all-reduced H and g (prior once)
    out = obj.fused(theta_ref, w[a:b])
    H_ref, jac_ref, S_ref = glm.sensitivity(family, X, y, theta_ref, w, prior=prior)
    assert rel(out["H"].cpu().numpy(), H_ref) < 1e-12
    g_ref = glm.grad(family, X, y, theta_ref, w, prior=prior)
    assert np.abs(out["g"].cpu().numpy() - g_ref).max() < 1e-9 * max(1.0, np.abs(H_ref).max())
    assert abs(obj(theta_ref, w[a:b]) - glm.value(family, X, y, theta_ref, w, prior=prior)) < 1e-9 * N

    # the drop-in class: S stays sharded like X
    la = va.HyperparameterSensitivityLinearApproximation(obj, theta_ref, w[a:b], validate_optimum=True,
                                                         grad_tol=1e-7)
    S = la.get_dopt_dhyper()
    assert S.shape == (D, b - a) and S.flags.f_contiguous
    assert rel(S, S_ref[:, a:b]) < 1e-10 * max(1.0, np.abs(S_ref).max() / np.abs(S_ref[:, a:b]).max())
    assert rel(la.get_hessian_at_opt(), H_ref) < 1e-12

    # predictions sum over the hyper axis: all-reduced, identical on every rank
    w_new = w * rng.uniform(0.8, 1.2, N)
    pred_ref = theta_ref + S_ref @ (w_new - w)
    assert rel(la.predict_opt_par_from_hyper_par(w_new[a:b]), pred_ref) < 1e-10
    Wn = np.stack([w_new, w, 2 * w - w_new], axis=1)          # (N, 3): three weight vectors as columns
    predb = la.predict_opt_par_from_hyper_par_batch(Wn[a:b])
    predb_ref = theta_ref[:, None] + S_ref @ (Wn - w[:, None])
    assert rel(predb, predb_ref) < 1e-10

    # differentiable optimum: the VJP is this rank's slice of the gradient, the JVP is all-reduced
    f = la.get_opt_par_function()
    wt = torch.tensor(w[a:b], device="cuda", requires_grad=True)
    gvec = rng.standard_normal(D)
    (f(wt) * torch.tensor(gvec, device="cuda")).sum().backward()
    assert rel(wt.grad.cpu().numpy(), (gvec @ S_ref)[a:b]) < 1e-10 * max(1.0, np.abs(gvec @ S_ref).max()
                                                                       / np.abs((gvec @ S_ref)[a:b]).max())
    tvec = rng.standard_normal(N)
    import torch.autograd.forward_ad as fwAD
    with fwAD.dual_level():
        dual = fwAD.make_dual(torch.tensor(w[a:b], device="cuda"), torch.tensor(tvec[a:b], device="cuda"))
        _, tang = fwAD.unpack_dual(f(dual))
    assert rel(tang.cpu().numpy(), S_ref @ tvec) < 1e-10

    # matrix-free path: all-reduced HVP, CG with replicated scalars
    hvp = GLMHessianVectorProduct(obj, theta_ref, w[a:b])
    v = rng.standard_normal(D)
    assert rel(hvp(v), glm.hvp(family, X, y, theta_ref, w, v, prior)) < 1e-10
    bvec = rng.standard_normal(D)
    mv = lambda u: glm.hvp(family, X, y, theta_ref, w, u, prior)
    x_ref, info_ref, it_ref = solvers.cg_legacy(mv, bvec, tol=1e-9, return_iters=True)
    xd, cinfo, iters = cg_solve(lambda p, o: hvp.apply(p, o), torch.tensor(bvec, device="cuda"), tol=1e-9)
    assert cinfo == info_ref == 0 and abs(iters - it_ref) <= 1
    assert rel(xd.cpu().numpy(), x_ref) < 1e-8
    # block CG on the sharded objective: K right-hand sides, ONE all-reduce of the K x D partial products per
    # iteration (multi-vector HVP kernels for K >= 4), every column against the unsharded dense solve
    Bk = rng.standard_normal((D, 6))
    Xk = va.solver_lib.get_cg_solver(hvp, D, cg_opts={'tol': 1e-12})(Bk)
    H_full = glm.hessian(family, X, y, theta_ref, w, prior)
    assert Xk.shape == (D, 6) and rel(Xk, np.linalg.solve(H_full, Bk)) < 1e-8, rel(Xk, np.linalg.solve(H_full, Bk))

    # data-weight Taylor terms up to order 4 (fused kernels to order 3, the Faa di Bruno route above): partial
    # right-hand sides all-reduced
    te = va.ParametricSensitivityTaylorExpansion.optimization_objective(obj, theta_ref, w[a:b], order=4)
    ds = te.evaluate_input_derivs((w_new - w)[a:b])
    ds_ref = taylor.input_derivs(family, X, y, theta_ref, w, w_new - w, prior=prior, max_order=4)
    for k in range(4):
        assert rel(ds[k], ds_ref[k]) < 1e-9, ("taylor", k, rel(ds[k], ds_ref[k]))
    # fp32 storage mode on the same shards (row n1): float X and S, all-reduced fp64 H, 1e-4
    if D <= 128:
        o32 = va.WeightedGLMObjective(X[a:b], y[a:b], family=family, prior=prior, process_group=group,
                                      dtype=torch.float32)
        la32 = va.HyperparameterSensitivityLinearApproximation(o32, theta_ref, w[a:b])
        S32 = la32.get_dopt_dhyper()
        assert S32.dtype == np.float32 and S32.shape == (D, b - a)
        assert rel(la32.get_hessian_at_opt(), H_ref) < 1e-4
        assert rel(S32, S_ref[:, a:b]) < 1e-4 * max(1.0, np.abs(S_ref).max() / np.abs(S_ref[:, a:b]).max())
        assert rel(la32.predict_opt_par_from_hyper_par(w_new[a:b]), pred_ref) < 1e-4
        th32, info32 = va.optim.newton_glm(o32, w[a:b], grad_tol=1e-9)
        assert info32["converged"] and rel(th32, theta_ref) < 1e-4
    return dict(family=family, N=N, D=D, rows=(a, b))


def vi_checks(rank, world, group):
    """Group-sharded hierarchical VI path: objective derivatives, block + global Hessian,
    the arrowhead solver with all-reduced Schur complement / global right-hand side, and the LR covariance,
    against the oracle on ALL groups."""
    import torch
    import vittles_amd as va
    from vittles_amd import sharding
    from oracle import vi, lr_cov
    G, ng, k = 37, 7, 3
    taus = (2.0, 1.5, 0.5)
    X, y = vi.synth(G, ng, k, 3)
    m = vi.HierGaussMFVI(X, y, G, ng, k, *taus)
    g0, g1 = sharding.shard_rows(G, rank, world)
    Gl = g1 - g0
    obj = va.HierGaussMFVIObjective(X[g0 * ng:g1 * ng], y[g0 * ng:g1 * ng], Gl, ng, k, *taus, process_group=group)
    nl, ngl = 2 * k * Gl, 2 * k

    def loc(vec):      # this rank's slice of a full-layout vector: its groups' entries + the globals
        return np.concatenate([vec[2 * k * g0:2 * k * g1], vec[2 * k * G:]])
    rng = np.random.default_rng(8)
    eta = m.optimum() * (1.0 + 0.03 * rng.random(m.d))
    w = 0.5 + rng.random(G * ng)
    u = rng.standard_normal(m.d)
    wl = w[g0 * ng:g1 * ng]
    assert abs(obj(loc(eta), wl) - m.value(eta, w)) <= 1e-11 * abs(m.value(eta, w))
    assert rel(obj.grad(loc(eta), wl).cpu().numpy(), loc(m.grad(eta, w))) < 1e-11
    assert rel(obj.hvp(loc(eta), loc(u), wl).cpu().numpy(), loc(m.hvp(eta, u, w))) < 1e-11

    eta0 = m.optimum()
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())
    H = sb.get_hessian(loc(eta0))
    Hd = m.hessian(eta0)
    sel = np.concatenate([np.arange(2 * k * g0, 2 * k * g1), np.arange(2 * k * G, m.d)])
    # the rank's matrix is H restricted to its (local + global) unknowns, with the FULL global corner
    assert rel(np.asarray(H.todense()), Hd[np.ix_(sel, sel)]) < 1e-11
    # the sharded solve == the matching rows of the global solve (Schur complement summed over ranks)
    b = rng.standard_normal((m.d, 2))
    x_ref = np.linalg.solve(Hd, b)
    solve = va.solver_lib.get_cholesky_solver(H)
    x = solve(b[sel])
    assert rel(x, x_ref[sel]) < 1e-10, ("arrow solve", rel(x, x_ref[sel]))
    # LR covariance of moments that live on different ranks and on the globals
    K = 6
    J = np.zeros((K, m.d))
    J[0, 0] = 1.0                      # m of group 0 (rank 0)
    J[1, 2 * k * (G - 1) + 1] = 1.0    # m of the last group (rank 1)
    J[2, 2 * k * G] = 1.0              # m_mu
    J[3, 2 * k * G + k] = 1.0          # v_mu
    J[4] = rng.standard_normal(m.d)    # dense moments
    J[5] = rng.standard_normal(m.d)
    lr = va.LinearResponseCovariances(obj, loc(eta0), validate_optimum=True, hessian_at_opt=H, grad_tol=1e-6)
    cov = lr.get_lr_covariance_from_jacobians(J[:, sel], J[:, sel])
    cov_ref = J @ np.linalg.solve(Hd, J.T)
    assert rel(cov, cov_ref) < 1e-10, ("lr cov", rel(cov, cov_ref))
    # off the optimum the Newton-step validation must fail on EVERY rank alike (all-reduced norm)
    try:
        va.LinearResponseCovariances(obj, loc(eta), validate_optimum=True, hessian_at_opt=H, grad_tol=1e-12)
        raise AssertionError("validation should have failed")
    except ValueError as e:
        assert "newton_step" in str(e)
    return dict(vi=True, groups=(g0, g1))


def main():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        group = dist.group.WORLD
        done = [glm_checks(rank, world, group, "logistic", 5001, 100),
                glm_checks(rank, world, group, "poisson", 1203, 24),
                glm_checks(rank, world, group, "logistic", 700, 200)]
        extra = globals().get("vi_checks")
        if extra is not None:
            done.append(extra(rank, world, group))
        print("rank %d ok: %s" % (rank, done), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
