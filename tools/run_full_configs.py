"""BASELINE.json configs[2] and configs[3] at FULL size on ONE MI355X (288 GB):
  c3: weighted logistic GLM, N = 10^7, D = 1000 (X 80 GB + S 80 GB): theta-hat by the device Newton
      optimiser, then Hessian -> Cholesky -> all-N sensitivities, timed; verified through
      size-independent properties (no CPU pass over 80 GB):
        * sampled rows: S[n] == -r_n H^{-1} x_n with SciPy's cho_solve on the host (H from the GPU),
        * H of the first rows == NumPy X^T D X on the host; H(all rows) == sum of H over 8 row shards,
        * checksum over ALL rows: sum_n w_n S[n, :] == -H^{-1} (g - prior theta-hat).
  c4: matrix-free CG, D = 20000 ridge-logistic, N = 5*10^5 (X 80 GB): directional sensitivity
      -H^{-1} X^T (r .* dw) at tol 1e-5 and 1e-12; HVP checked against NumPy on a row subset.
usage: python tools/run_full_configs.py [c3] [c4] [--scale F]   (F < 1 shrinks N for a rehearsal)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg
import torch

import vittles_amd as va
from vittles_amd import rows
from vittles_amd.linalg import DeviceCholesky

dev = torch.device("cuda", 0)


def note(msg):
    print("[%.1fs] %s" % (time.perf_counter() - T0, msg), flush=True)


SEED_C3 = 3


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def gen(N, D, seed):
    """The counter-based generator (vittles_amd/synth.py): the data set exists only in HBM, any row block can be
    regenerated on the host for the checks below."""
    X, y, _ = va.synth.glm_problem(N, D, "logistic", seed=seed, device=dev)
    return X, y


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    return out, e0.elapsed_time(e1)


def config3(scale):
    N, D = int(10_000_000 * scale), 1000
    note("c3: generating X (%d x %d, %.1f GB)" % (N, D, 8e-9 * N * D))
    X, y = gen(N, D, seed=SEED_C3)
    w = torch.ones(N, dtype=torch.float64, device=dev)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    t0 = time.perf_counter()
    theta, info = va.optim.newton_glm(obj, w, grad_tol=1e-9)
    torch.cuda.synchronize()
    t_newton = time.perf_counter() - t0
    note("c3: theta-hat in %d Newton iterations, %.2f s, |g| = %.2e" % (info["iterations"], t_newton, info["grad_norm"]))
    S = torch.empty((N, D), dtype=torch.float64, device=dev)
    res = {}
    for rep in range(2):                                   # second repetition is the reported one
        obj._cache_key = None
        out, ms_h = timed(lambda: obj.fused(theta, w))
        chol, ms_c = timed(lambda: DeviceCholesky.factor(out["H"], check=False, pack=True))
        _, ms_s = timed(lambda: chol.solve_rows(X, rowscale=out["r"], alpha=-1.0, out=S))
        chol.raise_if_failed()
        res = dict(hessian_ms=ms_h, potrf_pack_ms=ms_c, trsm_ms=ms_s, total_ms=ms_h + ms_c + ms_s)
    note("c3: hessian %.1f ms (%.1f TF triangle), potrf+pack %.1f ms, trsm %.1f ms (%.1f TF): %.3f s for all %d sensitivities"
         % (res["hessian_ms"], N * D * (D + 1) / res["hessian_ms"] / 1e9, res["potrf_pack_ms"], res["trsm_ms"],
            2.0 * N * D * D / res["trsm_ms"] / 1e9, res["total_ms"] / 1e3, N))
    res.update(N=N, D=D, newton_s=t_newton, newton_iterations=info["iterations"], grad_norm=info["grad_norm"],
               sens_per_s=N / (res["total_ms"] * 1e-3),
               syrk_TFLOPs_triangle=N * D * (D + 1) / res["hessian_ms"] / 1e9,
               trsm_TFLOPs=2.0 * N * D * D / res["trsm_ms"] / 1e9)
    # ---- checks -------------------------------------------------------------------------
    Hh = out["H"].cpu().numpy()
    gh = out["g"].cpu().numpy()
    th = theta.cpu().numpy()
    cf = scipy.linalg.cho_factor(Hh)
    idx = torch.randint(0, N, (2000,), device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    # the sampled rows are REGENERATED on the host from the counter-based seed, not copied from the device
    Xs, ys = va.synth.glm_rows_host(N, D, "logistic", SEED_C3, idx.cpu().numpy())
    res["regenerated_rows_max_abs_diff"] = float(np.abs(X[idx].cpu().numpy() - Xs).max())
    p = 1.0 / (1.0 + np.exp(-(Xs @ th)))
    S_ref = -scipy.linalg.cho_solve(cf, (Xs * (p - ys)[:, None]).T).T
    res["S_sampled_rows_rel_err"] = rel(S[idx].cpu().numpy(), S_ref)
    note("c3: 2000 sampled rows of S vs host cho_solve: %.2e" % res["S_sampled_rows_rel_err"])
    lhs = rows.gemv_t(S, w).cpu().numpy()
    rhs = -scipy.linalg.cho_solve(cf, gh - 1.0 * th)
    res["checksum_all_rows_rel_err"] = rel(lhs, rhs)
    note("c3: sum_n w_n S[n] vs -H^-1 (g - prior theta): %.2e" % res["checksum_all_rows_rel_err"])
    n_sub = min(N, 200_000)
    sub = va.WeightedGLMObjective(X[:n_sub], y[:n_sub], family="logistic", prior=1.0)
    H_sub = sub.fused(theta, w[:n_sub])["H"].cpu().numpy()
    Xh = X[:n_sub].cpu().numpy()
    ph = 1.0 / (1.0 + np.exp(-(Xh @ th)))
    H_ref = Xh.T @ ((ph * (1 - ph))[:, None] * Xh) + np.eye(D)
    res["H_first_rows_rel_err"] = rel(H_sub, H_ref)
    note("c3: H of the first %d rows vs NumPy: %.2e" % (n_sub, res["H_first_rows_rel_err"]))
    del Xh
    H_sum = np.zeros((D, D))
    for r in range(8):
        a, b = (N * r) // 8, (N * (r + 1)) // 8
        sh = va.WeightedGLMObjective(X[a:b], y[a:b], family="logistic", prior=1.0 if r == 0 else None)
        H_sum += sh.fused(theta, w[a:b])["H"].cpu().numpy()
    res["H_8_shards_rel_err"] = rel(H_sum, Hh)
    note("c3: sum of 8 row-shard Hessians vs the unsharded one: %.2e" % res["H_8_shards_rel_err"])
    return res


def config4(scale):
    N, D = int(500_000 * scale), 20_000
    note("c4: generating X (%d x %d, %.1f GB)" % (N, D, 8e-9 * N * D))
    X, y = gen(N, D, seed=4)
    w = torch.ones(N, dtype=torch.float64, device=dev)
    prior = 1.0
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior)
    from vittles_amd.cg import GLMHessianVectorProduct, cg_solve
    g = torch.Generator(device=dev).manual_seed(11)
    theta = torch.randn(D, dtype=torch.float64, device=dev, generator=g) / D ** 0.5
    hvp = GLMHessianVectorProduct(obj, theta, w)
    _, r, d = rows.glm_link(obj, theta, w)
    dw = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    b = rows.gemv_t2(X, r, dw)                                  # jac0 @ dw
    res = dict(N=N, D=D)
    v = torch.randn(D, dtype=torch.float64, device=dev, generator=g)
    outv = torch.empty_like(v)
    hvp.apply(v, outv)
    _, ms = timed(lambda: [hvp.apply(v, outv) for _ in range(5)])
    res["hvp_ms"] = ms / 5
    res["hvp_GBs"] = (8.0 * N * D + 8 * N + 16 * D) / (ms / 5) / 1e6
    note("c4: HVP %.2f ms = %.0f GB/s" % (res["hvp_ms"], res["hvp_GBs"]))
    for tol in (1e-5, 1e-12):
        t0 = time.perf_counter()
        x, info, iters = cg_solve(lambda p_, o_: hvp.apply(p_, o_), b, tol=tol)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        resid = torch.empty_like(x)
        hvp.apply(x, resid)
        rr = float((resid - b).norm() / b.norm())
        res["cg_tol_%g" % tol] = dict(iterations=iters, info=info, seconds=dt, ms_per_iteration=dt / max(iters, 1) * 1e3,
                                      true_residual=rr)
        note("c4: CG tol %g: %d iterations, %.2f s (%.2f ms/iteration), ||Hx-b||/||b|| = %.2e" %
             (tol, iters, dt, dt / max(iters, 1) * 1e3, rr))
    n_sub = min(N, 20_000)
    Xh = X[:n_sub].cpu().numpy()
    dh = d[:n_sub].cpu().numpy()
    vh = v.cpu().numpy()
    ref = Xh.T @ (dh * (Xh @ vh)) + prior * vh
    got = rows.glm_hvp(X[:n_sub], d[:n_sub], v, prior_scalar=prior).cpu().numpy()
    res["hvp_first_rows_rel_err"] = rel(got, ref)
    note("c4: HVP of the first %d rows vs NumPy: %.2e" % (n_sub, res["hvp_first_rows_rel_err"]))
    return res


if __name__ == "__main__":
    T0 = time.perf_counter()
    args = sys.argv[1:]
    scale = 1.0
    if "--scale" in args:
        scale = float(args[args.index("--scale") + 1])
    todo = [a for a in args if a in ("c3", "c4")] or ["c3", "c4"]
    out = {}
    if "c3" in todo:
        out["config3"] = config3(scale)
        torch.cuda.empty_cache()
    if "c4" in todo:
        out["config4"] = config4(scale)
    print(json.dumps(out))
