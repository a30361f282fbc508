#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X linear-response sensitivity hot path.

Workload (BASELINE.json configs[1]): weighted logistic regression, N = 1,000,000 rows,
D = 100, fp64, ridge prior 1: one "step" = all-N infinitesimal-jackknife sensitivities
d theta-hat / d w  from device-resident (X, y, w, theta-hat) to device-resident S:
    fused link+gradient+Hessian SYRK  ->  (all-reduce of [H | g] when row-sharded)
    -> Cholesky + TRSM operand pack -> MFMA multi-RHS TRSM  S = -H^{-1} (diag(r) X)^T
Metric: IJ sensitivities / second = (rows over all ranks) / (max-over-ranks step time).

Multi-GPU (weak scaling: every rank holds its own --rows shard; the only data-path collective
is the RCCL all-reduce of the (D*D + D)-double partial [H | g]):
  * under torchrun (WORLD_SIZE set) this process is one rank;
  * `python bench.py --gpus N` with N > 1 and no WORLD_SIZE starts N fresh rank processes
    itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE anything touches the GPU; the
    parent never initialises HIP and never execs.  VITTLES_BENCH_BACKEND=gloo lets the ranks
    share one card (rehearsal of the multi-rank code path on a one-GPU box).

Prints ONE JSON line (rank 0).
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X vendor FP64 matrix peak
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--rows", type=int, default=1_000_000, help="rows per GPU")
    p.add_argument("--dim", type=int, default=100)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-rows", type=int, default=1_000_000)
    p.add_argument("--no-hvp", action="store_true", help="skip the HVP GB/s + CG side measurement (configs[3])")
    p.add_argument("--hvp-rows", type=int, default=100_000)
    p.add_argument("--hvp-dim", type=int, default=20_000)
    p.add_argument("--cg-cpu-rows", type=int, default=4_000)
    p.add_argument("--no-block", action="store_true", help="skip the block-sparse VI side measurement (configs[4])")
    p.add_argument("--block-groups", type=int, default=10_000)
    p.add_argument("--no-fp32", action="store_true", help="skip the fp32 storage-mode side measurement (row n1)")
    p.add_argument("--no-c3", action="store_true", help="skip the configs[2] leg (D = 1000)")
    p.add_argument("--c3-rows", type=int, default=0,
                   help="TOTAL rows of the configs[2] leg (default: 1.25e6 on one GPU = an 8-GPU job's share, 1e7 over N > 1 GPUs)")
    p.add_argument("--c3-dim", type=int, default=1000)
    p.add_argument("--no-c4", action="store_true", help="skip the multi-rank configs[3] leg (N > 1 only)")
    p.add_argument("--c4-rows", type=int, default=500_000, help="TOTAL rows of the multi-rank configs[3] leg")
    return p.parse_args()


# ------------------------------------------------------------------------------------------------
# multi-rank launcher (parent process: no GPU call, no exec)
# ------------------------------------------------------------------------------------------------
def launch_ranks(n):
    """Start ``n`` fresh rank processes of this script and wait for them.  Runs before torch is
    imported in this process; rank 0's stdout (the JSON line) is this process's stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VITTLES_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        except OSError as e:
            for p in procs:
                p.kill()
            raise SystemExit("bench.py: could not start rank %d (%s); launch with "
                             "`python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d` instead"
                             % (r, e, n, n))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:          # one rank failed: the others would wait in a collective forever
                    q.terminate()
    if rc != 0:
        raise SystemExit("bench.py: a rank exited with code %d (world of %d); the torchrun route is "
                         "`python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                         "bench.py --gpus %d`" % (rc, n, n, n))


# ------------------------------------------------------------------------------------------------
# CPU baselines (oracle = checker and reported baseline only; never on the product path)
# ------------------------------------------------------------------------------------------------
def blas_threads():
    try:
        import threadpoolctl
        infos = threadpoolctl.threadpool_info()
        return int(max([i.get("num_threads", 1) for i in infos if i.get("user_api") == "blas"] or [1]))
    except Exception:
        return 1


def cpu_baseline(Xh, yh, wh, theta_h, prior, n_sample):
    """Time the oracle (CPU restatement: NumPy Hessian + the reference's LAPACK calls
    cho_factor/cho_solve with nrhs = N) on a bounded sample of the same workload and
    use it as the parity checker for the GPU result."""
    from oracle import glm
    n = min(n_sample, Xh.shape[0])
    Xs, ys, ws = Xh[:n], yh[:n], wh[:n]
    glm.sensitivity("logistic", Xs[:20000], ys[:20000], theta_h, ws[:20000], prior=prior)   # warm-up
    times = []
    for _ in range(5):      # warm-up 1, repeat >= 5, report the median
        t0 = time.perf_counter()
        H, jac0, S = glm.sensitivity("logistic", Xs, ys, theta_h, ws, prior=prior)
        times.append(time.perf_counter() - t0)
        if sum(times) > 30.0:
            break
    dt = float(np.median(times))
    return dict(value=n / dt, unit="sensitivities/s", cores=blas_threads(), os_cpu_count=os.cpu_count(),
                kind="port",
                sample="oracle.glm.sensitivity (NumPy X^T D X + scipy cho_factor/cho_solve, nrhs=N) on the first "
                       "%d rows of the same N x D workload, median of %d runs: %.2f s" % (n, len(times), dt)), H, S


def hvp_cg_side_measurement(torch, device, n_rows, dim, cpu_rows, reps=10):
    """BASELINE.json configs[3] (matrix-free CG, D = 20k ridge-logistic), after the timed region, N=1 only:
    (i) the metric's second figure: fused HVP kernel in algorithmic GB/s = (8ND + 8N + 16D) / kernel
        time, HIP events on the launch stream;
    (ii) one CG solve H x = b (tol 1e-5, the reference's default) through the drop-in get_cg_solver
        machinery, with its true residual;
    (iii) CPU baseline B3 in the same run: scipy.sparse.linalg.cg(rtol=1e-5, atol=0) over a NumPy HVP
        (solver_lib.py:91-97's LinearOperator) on the first ``cpu_rows`` rows."""
    import vittles_amd as va
    from vittles_amd import rows, cg
    gen = torch.Generator(device=device)
    gen.manual_seed(4321)
    X = torch.randn((n_rows, dim), dtype=torch.float64, device=device, generator=gen)
    theta_star = torch.randn(dim, dtype=torch.float64, device=device, generator=gen) / dim ** 0.5
    y = (torch.rand(n_rows, dtype=torch.float64, device=device, generator=gen) < torch.sigmoid(X @ theta_star)).double()
    w = torch.ones(n_rows, dtype=torch.float64, device=device)
    theta = 0.5 * theta_star
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    hvp = cg.GLMHessianVectorProduct(obj, theta, w)
    v = torch.randn(dim, dtype=torch.float64, device=device, generator=gen)
    out = torch.empty(dim, dtype=torch.float64, device=device)
    for _ in range(3):
        hvp.apply(v, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        hvp.apply(v, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 8.0 * n_rows * dim + 8.0 * n_rows + 16.0 * dim
    gbs = nbytes / (ms * 1e-3) / 1e9
    res_hvp = {"kernel": "k_rows_wide<OP_HVP> (K6 fused HVP)", "rows": n_rows, "dim": dim, "ms": ms,
               "algorithmic_bytes": nbytes, "GBs": gbs, "peak_GBs": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}

    # (ii) CG: right-hand side = the cross-Hessian applied to a weight perturbation, X^T (r .* dw)
    _, r, _ = rows.glm_link(obj, theta, w)
    dw = torch.randn(n_rows, dtype=torch.float64, device=device, generator=gen)
    b = rows.gemv_t2(X, r, dw)
    tol = 1e-5
    cg.cg_solve(lambda p, q: hvp.apply(p, q), b, tol=tol)          # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        x, info, iters = cg.cg_solve(lambda p, q: hvp.apply(p, q), b, tol=tol)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    gpu_s = float(np.median(times))
    resid = float(((hvp.apply(x) - b).norm() / b.norm()).item())

    # (ii-b) block CG: K = 16 right-hand sides in lock-step, one multi-vector HVP (two MFMA passes over X) per
    # iteration -- 16 sensitivity directions for about the X traffic of 3 single solves
    Kb = 16
    dW = torch.randn((Kb, n_rows), dtype=torch.float64, device=device, generator=gen)
    Bk = torch.stack([rows.gemv_t2(X, r, dW[k]) for k in range(Kb)])
    cg.cg_solve_multi(hvp.apply_multi, Bk, tol=tol, maxiter=2)        # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    Xk, infos, its = cg.cg_solve_multi(hvp.apply_multi, Bk, tol=tol)
    torch.cuda.synchronize()
    blk_s = time.perf_counter() - t0
    Rk = hvp.apply_multi(Xk) - Bk
    blk_resid = float((Rk.norm(dim=1) / Bk.norm(dim=1)).max().item())
    res_blk = {"columns": Kb, "gpu_ms": blk_s * 1e3, "iterations_max": int(its.max()), "iterations_min": int(its.min()),
               "info_nonzero": int((infos != 0).sum()), "true_residual_rel_max": blk_resid,
               "ms_per_column": blk_s * 1e3 / Kb, "single_solve_ms": gpu_s * 1e3,
               "speedup_vs_columns_one_by_one": Kb * gpu_s / blk_s}
    del dW, Bk, Xk, Rk

    # (ii-c) the optional fp32 STORAGE mode on the same rows: X rounded to float (half the bytes per product),
    # vectors and every sum fp64 -- the wide row kernel on 16-byte lanes of four floats
    obj32 = va.WeightedGLMObjective(X.to(torch.float32), y, family="logistic", prior=1.0, dtype=torch.float32)
    hvp32 = cg.GLMHessianVectorProduct(obj32, theta, w)
    _, r32, _ = rows.glm_link(obj32, theta, w)
    b32 = rows.gemv_t2(obj32.X, r32, dw)
    cg.cg_solve(lambda p, q: hvp32.apply(p, q), b32, tol=tol)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        hvp32.apply(v, out)
    e1.record()
    torch.cuda.synchronize()
    ms32 = e0.elapsed_time(e1) / reps
    times32 = []
    for _ in range(3):
        t0 = time.perf_counter()
        x32, info32, iters32 = cg.cg_solve(lambda p, q: hvp32.apply(p, q), b32, tol=tol)
        torch.cuda.synchronize()
        times32.append(time.perf_counter() - t0)
    res_f32 = {"note": "optional mode, never the headline: X stored in float, vectors and every sum fp64",
               "hvp_ms": ms32, "hvp_GBs": (4.0 * n_rows * dim + 8.0 * n_rows + 16.0 * dim) / (ms32 * 1e-3) / 1e9,
               "gpu_ms": float(np.median(times32)) * 1e3, "gpu_iterations": int(iters32), "gpu_info": int(info32),
               "solution_rel_diff_vs_fp64": float(((x32 - x).norm() / x.norm()).item()),
               "speedup_vs_fp64": gpu_s / float(np.median(times32))}
    del obj32, hvp32

    # (iii) CPU B3 on a bounded row sample (same D, same tolerance)
    import scipy.sparse.linalg as spla
    from oracle import glm
    nc = min(cpu_rows, n_rows)
    Xc = X[:nc].cpu().numpy()
    yc = y[:nc].cpu().numpy()
    wc = np.ones(nc)
    th = theta.cpu().numpy()
    _, l1, l2 = glm.link("logistic", Xc @ th, yc)
    bc = Xc.T @ (l1 * dw[:nc].cpu().numpy())
    n_mv = [0]

    def mv(vv):
        n_mv[0] += 1
        return Xc.T @ (l2 * (Xc @ vv)) + vv
    linop = spla.LinearOperator((dim, dim), mv)
    t0 = time.perf_counter()
    xc, cinfo = spla.cg(linop, bc, rtol=tol, atol=0.0)
    cpu_s = time.perf_counter() - t0
    cpu_iters = n_mv[0]
    del X
    res_cg = {"workload": "matrix-free CG, ridge-logistic D=%d, N=%d rows on the GPU (configs[3]), tol=%g" % (dim, n_rows, tol),
              "gpu_ms": gpu_s * 1e3, "gpu_iterations": int(iters), "gpu_info": int(info),
              "gpu_true_residual_rel": resid,
              "gpu_row_iterations_per_s": n_rows * max(iters, 1) / gpu_s,
              "cpu_baseline": {"kind": "port", "what": "scipy.sparse.linalg.cg(rtol=%g, atol=0) + NumPy HVP "
                               "(X^T(d*(Xv)) + v) on the first %d rows, same D" % (tol, nc),
                               "rows": nc, "ms": cpu_s * 1e3, "iterations": cpu_iters, "info": int(cinfo),
                               "cores": blas_threads(), "os_cpu_count": os.cpu_count(),
                               "row_iterations_per_s": nc * max(cpu_iters, 1) / cpu_s}}
    res_cg["speedup_row_iterations"] = res_cg["gpu_row_iterations_per_s"] / res_cg["cpu_baseline"]["row_iterations_per_s"]
    res_cg["block_cg"] = res_blk
    res_cg["fp32_mode"] = res_f32
    return res_hvp, res_cg


def block_side_measurement(torch, G, ng=16, k=4):
    """BASELINE.json configs[4] (hierarchical mean-field Gaussian VI, G groups): SparseBlockHessian
    assembly + LinearResponseCovariances through the drop-in classes (host arrays in / out, like the
    reference's API), and CPU baseline B4 in the same run: the oracle's restatement of the reference
    loops (sparse_hessian_lib.py:93-102, 130-156) over a NumPy HVP + scipy.sparse.linalg.factorized
    (SuperLU, solver_lib.py:48) + J @ solve(J^T) (lr_cov_lib.py:172)."""
    import vittles_amd as va
    from oracle import vi, block_hessian, lr_cov
    X, y = vi.synth(G, ng, k, 0)
    m = vi.HierGaussMFVI(X, y, G, ng, k, 2.0, 1.5, 0.5)
    eta = m.optimum()
    nm = 8
    K = nm * k + k
    J = np.zeros((K, m.d))
    for g in range(nm):
        for j in range(k):
            J[g * k + j, 2 * k * g + j] = 1.0
    for j in range(k):
        J[nm * k + j, 2 * k * G + j] = 1.0

    obj = va.HierGaussMFVIObjective(X, y, G, ng, k, 2.0, 1.5, 0.5)
    sb = va.SparseBlockHessian(obj, obj.sparsity_array())

    def gpu_all():
        t0 = time.perf_counter()
        H = sb.get_hessian(eta)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lr = va.LinearResponseCovariances(obj, eta, validate_optimum=True, hessian_at_opt=H, grad_tol=1e-6)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        cov = lr.get_lr_covariance_from_jacobians(J, J)
        t3 = time.perf_counter()
        return (t1 - t0, t2 - t1, t3 - t2), H, cov
    gpu_all()
    runs = [gpu_all() for _ in range(3)]
    tg = np.median(np.array([r[0] for r in runs]), axis=0)
    Hg, cov_g = runs[-1][1], runs[-1][2]

    t0 = time.perf_counter()
    Hc = block_hessian.SparseBlockHessian(m.hvp, m.sparsity_array()).get_hessian(eta)
    t1 = time.perf_counter()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # SparseEfficiencyWarning: COO -> CSC, as in the reference
        lrc = lr_cov.LinearResponseCovariances(m.grad, None, eta, validate_optimum=True, hessian_at_opt=Hc,
                                               grad_tol=1e-6)
        t2 = time.perf_counter()
        cov_c = lrc.get_lr_covariance_from_jacobians(J, J)
    t3 = time.perf_counter()
    dH = abs(Hg - Hc)
    return {"workload": "hierarchical MF-Gaussian VI, G=%d groups x %d rows, k=%d (d=%d), %d moments (configs[4])"
                        % (G, ng, k, m.d, K),
            "gpu_ms": {"get_hessian_host_to_host": tg[0] * 1e3, "lr_setup_factor_validate": tg[1] * 1e3,
                       "lr_covariance": tg[2] * 1e3, "total": float(tg.sum()) * 1e3},
            "cpu_baseline": {"kind": "port", "what": "oracle SparseBlockHessian loops over a NumPy HVP + "
                             "scipy.sparse.linalg.factorized (SuperLU) + J @ solve(J^T)",
                             "ms": {"get_hessian": (t1 - t0) * 1e3, "lr_setup_factor_validate": (t2 - t1) * 1e3,
                                    "lr_covariance": (t3 - t2) * 1e3, "total": (t3 - t0) * 1e3},
                             "cores": 1, "os_cpu_count": os.cpu_count()},
            "speedup_total": (t3 - t0) / float(tg.sum()),
            "parity": {"hessian_max_rel_err": float(dH.max() / abs(Hc).max()), "nnz": [int(Hg.nnz), int(Hc.nnz)],
                       "lr_cov_max_rel_err": float(np.abs(cov_g - cov_c).max() / np.abs(cov_c).max()),
                       "tolerance": 1e-10}}


def config1_side_measurement(torch):
    """BASELINE.json configs[0] (N = 10k, D = 20, the reference's own CPU-runnable case), both ways:
    the drop-in class on the GPU (host arrays in, host S out: construction to get_dopt_dhyper)
    and an AD-structured CPU baseline with the reference's cost structure -- torch.func.hessian
    + jacrev(grad, argnums=1) in fp64 (D + D reverse sweeps) + SciPy cho_factor / cho_solve --
    standing in for `vittles` + `autograd`, which is not installed."""
    import scipy.linalg
    import vittles_amd as va
    from oracle import glm
    N, D = 10_000, 20
    X, y, w, _ = glm.synth(N, D, "logistic", seed=0)
    theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)

    def gpu():
        obj.invalidate_cache()
        la = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
        return la.get_dopt_dhyper()
    gpu()
    t0 = time.perf_counter()
    for _ in range(5):
        S_gpu = gpu()
    gpu_ms = (time.perf_counter() - t0) / 5 * 1e3
    Xt, yt = torch.tensor(X), torch.tensor(y)

    def f(t, ww):
        z = Xt @ t
        return (ww * (torch.nn.functional.softplus(z) - yt * z)).sum() + 0.5 * (t * t).sum()

    def cpu_ad():
        tt, wt = torch.tensor(theta), torch.tensor(w)
        H = torch.func.hessian(f, 0)(tt, wt).numpy()
        jac0 = torch.func.jacrev(torch.func.grad(f, 0), 1)(tt, wt).numpy()
        return -1 * scipy.linalg.cho_solve(scipy.linalg.cho_factor(H), jac0)
    cpu_ad()
    t0 = time.perf_counter()
    for _ in range(3):
        S_ad = cpu_ad()
    ad_ms = (time.perf_counter() - t0) / 3 * 1e3
    err = float(np.abs(S_gpu - S_ad).max() / np.abs(S_ad).max())
    return {"workload": "weighted logistic regression N=10000 D=20 (configs[0])",
            "gpu_dropin_ms_host_to_host": gpu_ms, "gpu_sens_per_s": N / (gpu_ms * 1e-3),
            "cpu_ad_baseline_ms": ad_ms, "cpu_ad_sens_per_s": N / (ad_ms * 1e-3),
            "cpu_ad_kind": "torch.func.hessian + jacrev(grad, 1) fp64 + scipy cho_factor/cho_solve (stand-in for "
                           "vittles+autograd, not installed)", "cpu_threads": torch.get_num_threads(),
            "os_cpu_count": os.cpu_count(), "S_max_rel_err_gpu_vs_ad": err}


FP32_MFMA_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense


def fp32_side_measurement(torch, va, DeviceCholesky, X, y, w, theta, prior, H64, S64, reps=200):
    """the same pipeline with X and S stored in float (optional mode, tolerance
    1e-4 against fp64, NEVER the headline `value`).  Parity is taken against this run's fp64 H and S."""
    N, D = X.shape
    X32 = X.to(torch.float32)
    obj = va.WeightedGLMObjective(X32, y, family="logistic", prior=prior, dtype=torch.float32)
    S32 = torch.empty((N, D), dtype=torch.float32, device=X.device)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps)]

    def step(i=None):
        obj.invalidate_cache()
        if i is not None:
            ev[i][0].record()
        out = obj.fused(theta, w)
        if i is not None:
            ev[i][1].record()
        chol = DeviceCholesky.factor(out["H"], check=False, pack=True)
        chol.pack32()
        if i is not None:
            ev[i][2].record()
        chol.solve_rows(X32, rowscale=out["r"], alpha=-1.0, out=S32)
        if i is not None:
            ev[i][3].record()
        return out
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        out = step(i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    ms_h = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_c = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    ms_t = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
    res = {"note": "optional fp32 STORAGE mode (X and S float; link, r, d, H, g, Cholesky double): reduced precision, "
                   "reported beside the fp64 headline, never as `value`",
           "dtype": "f32 storage / f32 MFMA products / fp64 sums", "ms_per_step": ms,
           "sensitivities_per_s": N / (ms * 1e-3),
           "stages_ms": {"fused_link_grad_hessian": ms_h, "cholesky_pack": ms_c, "sens_trsm": ms_t},
           "hessian_TFLOPs_triangle": float(N) * D * (D + 1) / (ms_h * 1e-3) / 1e12,
           "trsm_TFLOPs": 2.0 * D * D * N / (ms_t * 1e-3) / 1e12,
           "trsm_hbm_GBs": (8.0 * N * D + 8.0 * N) / (ms_t * 1e-3) / 1e9,
           "fp32_mfma_peak_TFLOPs": FP32_MFMA_PEAK_TFLOPS,
           "tolerance": 1e-4,
           "H_max_rel_err_vs_fp64": float(((out["H"] - H64).abs().max() / H64.abs().max()).item())}
    err = 0.0
    smax = float(S64.abs().max().item())
    for a in range(0, N, 1 << 18):     # in slabs: no second N x D fp64 temporary
        err = max(err, float((S32[a:a + (1 << 18)].double() - S64[a:a + (1 << 18)]).abs().max().item()))
    res["S_max_rel_err_vs_fp64"] = err / smax
    return res


def _max_over_ranks(torch, dist, world, coll_dev, x):
    t = torch.tensor([float(x)], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def config3_leg(torch, dist, va, DeviceCholesky, world, rank, device, group, coll_dev, rows_total, dim, reps=3):
    """BASELINE.json configs[2] (weighted logistic GLM, D = 1000; N = 10^7 over the whole job), rank-symmetric:
    every rank holds rows_total / world rows of ONE Philox data set, theta-hat by the device Newton optimiser,
    then Hessian (+ all-reduce of [H | g]) -> Cholesky + pack -> all-rows TRSM, timed per stage with HIP events
    and as a whole (barrier + max over ranks).  At world == 1 the default is the 1/8 share an 8-GPU job gives one
    card (1.25e6 rows); at world > 1 it is the strong-scaled 10^7 rows.  Parity through size-independent
    properties: sampled rows of S regenerated on the host against SciPy's cho_solve, the all-rows checksum
    sum_n w_n S[n, :] == -H^{-1} (g - prior theta), H of the first rows against NumPy, and H identical on all
    ranks after the all-reduce."""
    import scipy.linalg
    from vittles_amd import rows as vrows
    seed = 3
    n_local = rows_total // world
    N = n_local * world
    X, y, _ = va.synth.glm_problem(n_local, dim, "logistic", seed=seed, row0=rank * n_local, device=device)
    w = torch.ones(n_local, dtype=torch.float64, device=device)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0, process_group=group)
    t0 = time.perf_counter()
    theta, info = va.optim.newton_glm(obj, w, grad_tol=1e-9)
    torch.cuda.synchronize()
    t_newton = time.perf_counter() - t0
    S = torch.empty((n_local, dim), dtype=torch.float64, device=device)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(reps)]
    cur = [None]

    def hook(name):
        if cur[0] is not None:
            ev[cur[0]][4 if name == "allreduce_begin" else 5].record()
    obj.stage_hook = hook if world > 1 else None

    def step(i=None):
        obj.invalidate_cache()
        cur[0] = i
        if i is not None:
            ev[i][0].record()
        out = obj.fused(theta, w)
        if i is not None:
            ev[i][1].record()
        chol = DeviceCholesky.factor(out["H"], check=False, pack=True)
        if i is not None:
            ev[i][2].record()
        chol.solve_rows(X, rowscale=out["r"], alpha=-1.0, out=S)
        if i is not None:
            ev[i][3].record()
        chol.raise_if_failed()
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    step()
    barrier()
    t0 = time.perf_counter()
    for i in range(reps):
        out = step(i)
    barrier()
    dt = _max_over_ranks(torch, dist, world, coll_dev, (time.perf_counter() - t0) / reps)
    cur[0] = None
    obj.stage_hook = None
    ms_h = float(np.median([e[0].elapsed_time(e[1]) for e in ev]))
    ms_c = float(np.median([e[1].elapsed_time(e[2]) for e in ev]))
    ms_t = float(np.median([e[2].elapsed_time(e[3]) for e in ev]))
    ms_ar = float(np.median([e[4].elapsed_time(e[5]) for e in ev])) if world > 1 else 0.0
    ms_syrk = ms_h - ms_ar
    # ---- parity ----
    Hh = out["H"].cpu().numpy()
    gh = out["g"].cpu().numpy()
    th = theta.cpu().numpy()
    cf = scipy.linalg.cho_factor(Hh)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    idx = np.unique(np.linspace(0, n_local - 1, 500).astype(np.int64))
    Xs, ys = va.synth.glm_rows_host(N, dim, "logistic", seed, idx + rank * n_local)
    di = torch.as_tensor(idx, device=device)
    regen = float(np.abs(X[di].cpu().numpy() - Xs).max())
    pr = 1.0 / (1.0 + np.exp(-(Xs @ th)))
    S_ref = -scipy.linalg.cho_solve(cf, (Xs * (pr - ys)[:, None]).T).T
    err_rows = _max_over_ranks(torch, dist, world, coll_dev, rel(S[di].cpu().numpy(), S_ref))
    lhs = vrows.gemv_t(S, w)
    obj._all_reduce(lhs)
    err_sum = rel(lhs.cpu().numpy(), -scipy.linalg.cho_solve(cf, gh - 1.0 * th))
    hs = torch.tensor([float(out["H"].sum().item()), float(out["H"].abs().sum().item())], dtype=torch.float64,
                      device=coll_dev)
    same = True
    if world > 1:
        allh = [torch.zeros_like(hs) for _ in range(world)]
        dist.all_gather(allh, hs)
        same = all(bool(torch.equal(a, allh[0])) for a in allh)
    res = None
    if rank == 0:
        n_sub = min(n_local, 20_000)
        sub = va.WeightedGLMObjective(X[:n_sub], y[:n_sub], family="logistic", prior=1.0)
        H_sub = sub.fused(theta, w[:n_sub])["H"].cpu().numpy()
        Xh = X[:n_sub].cpu().numpy()
        ph = 1.0 / (1.0 + np.exp(-(Xh @ th)))
        H_ref = Xh.T @ ((ph * (1 - ph))[:, None] * Xh) + np.eye(dim)
        syrk_tf = float(n_local) * dim * (dim + 1) / (ms_syrk * 1e-3) / 1e12
        trsm_tf = 2.0 * dim * dim * n_local / (ms_t * 1e-3) / 1e12
        res = {"workload": "weighted logistic GLM D=%d, %d rows per GPU x %d GPU(s) = %d rows (configs[2]%s), fp64"
                           % (dim, n_local, world, N, "" if N >= 10_000_000 else ": the share one of 8 GPUs holds"
                              if N == 1_250_000 and world == 1 else ", reduced rows"),
               "rows_per_gpu": n_local, "rows_total": N, "dim": dim, "n_gpus": world,
               "scaling": "strong" if world > 1 else "one share",
               "stages_ms": {"fused_link_grad_hessian": ms_syrk, "all_reduce_H_g": ms_ar, "cholesky_pack": ms_c,
                             "sens_trsm": ms_t},
               "seconds_all_sensitivities": dt, "sensitivities_per_s": N / dt,
               "syrk_TFLOPs": syrk_tf, "syrk_frac_of_peak": syrk_tf / FP64_MFMA_PEAK_TFLOPS,
               "trsm_TFLOPs": trsm_tf, "trsm_frac_of_peak": trsm_tf / FP64_MFMA_PEAK_TFLOPS,
               "flops_note": "per GPU; SYRK = N D (D+1) (the triangle), TRSM = 2 D^2 N; peak %.1f TFLOP/s fp64 MFMA"
                             % FP64_MFMA_PEAK_TFLOPS,
               "newton": {"seconds": t_newton, "iterations": int(info["iterations"]),
                          "grad_norm": float(info["grad_norm"])},
               "parity": {"S_sampled_rows_rel_err_max_over_ranks": err_rows, "sampled_rows_per_rank": int(len(idx)),
                          "regenerated_rows_max_abs_diff": regen,
                          "checksum_all_rows_rel_err": err_sum,
                          "H_first_rows_rel_err_vs_numpy": rel(H_sub, H_ref), "H_first_rows": int(n_sub),
                          "H_identical_on_all_ranks": bool(same), "tolerance": 1e-9}}
    if world == 1 and rank == 0 and dim <= 4096:
        # the optional fp32 STORAGE mode at this D (row n1): X and S float (half the footprint), fp64 products on
        # float panels, slab-wise fp64 solve; reported beside the fp64 figures, never as `value`
        S64_rows = S[di].clone()
        del S
        X32 = X.to(torch.float32)
        obj32 = va.WeightedGLMObjective(X32, y, family="logistic", prior=1.0, dtype=torch.float32)
        S32 = torch.empty((n_local, dim), dtype=torch.float32, device=device)
        ev32 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        for rep in range(2):
            obj32.invalidate_cache()
            ev32[0].record()
            o32 = obj32.fused(theta, w)
            ev32[1].record()
            c32 = DeviceCholesky.factor(o32["H"], check=False, pack=True)
            ev32[2].record()
            c32.solve_rows(obj32.X, rowscale=o32["r"], alpha=-1.0, out=S32)
            ev32[3].record()
            c32.raise_if_failed()
        torch.cuda.synchronize()
        t32 = [ev32[i].elapsed_time(ev32[i + 1]) for i in range(3)]
        res["fp32_mode"] = {
            "note": "optional fp32 STORAGE mode at D=%d: X and S float (%.1f GB each instead of %.1f), link / r / d / H / "
                    "g / Cholesky double, SYRK on float panels with fp64 products, solve in fp64 row slabs; reduced "
                    "precision, never `value`" % (dim, 4e-9 * n_local * dim, 8e-9 * n_local * dim),
            "stages_ms": {"fused_link_grad_hessian": t32[0], "cholesky_pack": t32[1], "sens_trsm": t32[2]},
            "sensitivities_per_s": n_local / (sum(t32) * 1e-3),
            "syrk_TFLOPs": float(n_local) * dim * (dim + 1) / (t32[0] * 1e-3) / 1e12,
            "trsm_TFLOPs": 2.0 * dim * dim * n_local / (t32[2] * 1e-3) / 1e12,
            "tolerance": 1e-4,
            "H_max_rel_err_vs_fp64": float(((o32["H"] - out["H"]).abs().max() / out["H"].abs().max()).item()),
            "S_sampled_rows_rel_err_vs_fp64": float(((S32[di].double() - S64_rows).abs().max()
                                                     / S64_rows.abs().max()).item())}
        del X32, obj32, S32, o32, c32
        S = None
    del X, S, obj
    torch.cuda.empty_cache()
    return res


def config4_leg(torch, dist, va, world, rank, device, group, coll_dev, rows_total, dim, tol=1e-5):
    """BASELINE.json configs[3] (matrix-free CG, D = 20000 ridge-logistic, N = 5*10^5 over the whole job),
    rank-symmetric: row-sharded fused HVP + one all-reduce of the D-vector per CG iteration.  Reports the solve,
    the local HVP kernel and the collective timed on its own."""
    from vittles_amd import rows as vrows, cg
    n_local = rows_total // world
    N = n_local * world
    X, y, theta_star = va.synth.glm_problem(n_local, dim, "logistic", seed=4, row0=rank * n_local, device=device)
    w = torch.ones(n_local, dtype=torch.float64, device=device)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0, process_group=group)
    theta = 0.5 * theta_star
    hvp = cg.GLMHessianVectorProduct(obj, theta, w)
    _, r, _ = vrows.glm_link(obj, theta, w)
    dw = va.synth.normal_rows(5, rank * n_local, n_local, 2, device=device)[:, 0].contiguous()
    b = vrows.gemv_t2(X, r, dw)
    obj._all_reduce(b)
    v = torch.as_tensor(va.synth.theta_star_host(6, dim), device=device)
    out = torch.empty_like(v)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    ms_local = timed(lambda: vrows.glm_hvp(obj.X, hvp.d, v, prior=None, out=out), 10)
    ms_ar = timed(lambda: obj._all_reduce(out), 20)
    ms_apply = timed(lambda: hvp.apply(v, out), 10)
    cg.cg_solve(lambda p_, o_: hvp.apply(p_, o_), b, tol=tol, maxiter=3)     # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x, info, iters = cg.cg_solve(lambda p_, o_: hvp.apply(p_, o_), b, tol=tol)
    torch.cuda.synchronize()
    dt = _max_over_ranks(torch, dist, world, coll_dev, time.perf_counter() - t0)
    resid = float(((hvp.apply(x) - b).norm() / b.norm()).item())
    res = None
    if rank == 0:
        nbytes = 8.0 * n_local * dim + 8.0 * n_local + 16.0 * dim
        res = {"workload": "matrix-free CG, ridge-logistic D=%d, %d rows per GPU x %d GPU(s) = %d rows (configs[3]), "
                           "tol=%g" % (dim, n_local, world, N, tol),
               "rows_per_gpu": n_local, "rows_total": N, "dim": dim, "n_gpus": world, "scaling": "strong",
               "cg_seconds": dt, "cg_iterations": int(iters), "cg_info": int(info), "true_residual_rel": resid,
               "ms_per_iteration": dt / max(iters, 1) * 1e3,
               "hvp_local_kernel_ms": ms_local, "hvp_local_GBs": nbytes / (ms_local * 1e-3) / 1e9,
               "hvp_local_frac_of_hbm_peak": nbytes / (ms_local * 1e-3) / 1e9 / HBM_PEAK_GBS,
               "all_reduce_D_vector_ms": ms_ar, "all_reduce_bytes": 8 * dim,
               "hvp_with_all_reduce_ms": ms_apply,
               "row_iterations_per_s": N * max(iters, 1) / dt}
    del X, obj, hvp
    torch.cuda.empty_cache()
    # the optional fp32 STORAGE mode on the same rows: X kept in float (half the bytes per product), every sum fp64
    X32, y32, _ = va.synth.glm_problem(n_local, dim, "logistic", seed=4, row0=rank * n_local, device=device,
                                       dtype=torch.float32)
    obj32 = va.WeightedGLMObjective(X32, y32, family="logistic", prior=1.0, process_group=group,
                                    dtype=torch.float32)
    hvp32 = cg.GLMHessianVectorProduct(obj32, theta, w)
    _, r32, _ = vrows.glm_link(obj32, theta, w)
    b32 = vrows.gemv_t2(obj32.X, r32, dw)
    obj32._all_reduce(b32)
    ms_local32 = timed(lambda: vrows.glm_hvp(obj32.X, hvp32.d, v, prior=None, out=out), 10)
    cg.cg_solve(lambda p_, o_: hvp32.apply(p_, o_), b32, tol=tol, maxiter=3)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x32, info32, iters32 = cg.cg_solve(lambda p_, o_: hvp32.apply(p_, o_), b32, tol=tol)
    torch.cuda.synchronize()
    dt32 = _max_over_ranks(torch, dist, world, coll_dev, time.perf_counter() - t0)
    if rank == 0:
        res["fp32_mode"] = {
            "note": "X stored in float (rows of 4 * D bytes), vectors and every sum fp64; optional mode, never the headline",
            "cg_seconds": dt32, "cg_iterations": int(iters32), "cg_info": int(info32),
            "hvp_local_kernel_ms": ms_local32,
            "hvp_local_GBs": (4.0 * n_local * dim + 8.0 * n_local + 16.0 * dim) / (ms_local32 * 1e-3) / 1e9,
            "solution_rel_diff_vs_fp64": float(((x32 - x).norm() / x.norm()).item()),
            "speedup_vs_fp64": dt / dt32}
    del X32, obj32, hvp32
    torch.cuda.empty_cache()
    return res



def note(msg):
    if os.environ.get("VITTLES_BENCH_VERBOSE"):
        print("[bench rank %s %.1fs] %s" % (os.environ.get("RANK", "0"), time.perf_counter(), msg),
              file=sys.stderr, flush=True)


def latest_traffic(N, D):
    """HBM traffic of the dominant kernel: PMC counters can only be collected under rocprofv3
    (separate --pmc passes), so the per-launch figure measured for this exact workload is read from
    the newest profiles/rNN_traffic.json and reported as null for any other shape."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            tj = json.load(open(path))
            if tj["workload"] == {"rows": N, "dim": D}:
                k = tj["k_trsm_rows_small"]
                return k["hbm_bytes_per_launch"], "%s (%s)" % (os.path.relpath(path, ROOT),
                                                               tj.get("command", "rocprofv3 --pmc passes, see file"))
        except Exception:
            continue
    return None, None


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)          # parent: never touches the GPU
        return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # one rank per GPU; VITTLES_BENCH_BACKEND=gloo lets several ranks share one card to
    # rehearse the multi-rank code path on a one-GPU box (RCCL refuses duplicate devices)
    backend = os.environ.get("VITTLES_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if world > ndev and backend == "nccl":
        raise SystemExit("bench.py: %d ranks but only %d GPU(s) visible; RCCL needs one GPU per rank "
                         "(VITTLES_BENCH_BACKEND=gloo rehearses the multi-rank path on one card)" % (world, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD
    coll_dev = device if backend == "nccl" else "cpu"

    note("process group up")
    import vittles_amd as va
    from vittles_amd.linalg import DeviceCholesky

    N, D = args.rows, args.dim
    prior = 1.0
    # SURVEY.md section 8d synthetic inputs, generated in HBM by the counter-based generator (Philox-4x32-10, seed 0):
    # rank r holds rows r*N .. (r+1)*N - 1 of ONE global data set, whatever the number of ranks is
    X, y, theta_star = va.synth.glm_problem(N, D, "logistic", seed=0, row0=rank * N, device=device)
    w = torch.ones(N, dtype=torch.float64, device=device)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior, process_group=group)

    # theta-hat: damped Newton on the device with the same kernels (set-up, untimed).
    theta, _info = va.optim.newton_glm(obj, w, grad_tol=1e-10)
    grad_norm = float(obj.fused(theta.clone(), w)["g"].norm())
    note("theta-hat found, |g| = %.2e" % grad_norm)

    S = torch.empty((N, D), dtype=torch.float64, device=device)
    nev = 6
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(nev)] for _ in range(args.steps)]
    cur = [None]

    def hook(name):          # events around the all-reduce inside obj.fused (timed steps only)
        if cur[0] is not None:
            ev[cur[0]][4 if name == "allreduce_begin" else 5].record()
    obj.stage_hook = hook if world > 1 else None

    # potrf's LAPACK-style info of every step is read on the host inside the timed region, one
    # step late (pinned buffer + event), so the check never drains the stream
    info_host = torch.zeros(max(args.steps, 1), dtype=torch.int32).pin_memory()
    info_ev = [None] * args.steps

    def check_info(i):
        info_ev[i].synchronize()
        DeviceCholesky.raise_for_info(int(info_host[i]))

    def step(i=None):
        obj.invalidate_cache()              # no cached pass: every step streams X
        cur[0] = i
        if i is not None:
            ev[i][0].record()
        out = obj.fused(theta, w)           # K0+K1+K3 (+ all-reduce)
        if i is not None:
            ev[i][1].record()
        chol = DeviceCholesky.factor(out["H"], check=False, pack=True)   # K4 + TRSM pack, one launch
        if i is not None:
            ev[i][2].record()
        chol.solve_rows(X, rowscale=out["r"], alpha=-1.0, out=S)   # K2+K5
        if i is not None:
            ev[i][3].record()
            info_ev[i] = chol.info_to_host_async(info_host[i:i + 1])
            if i > 0:
                check_info(i - 1)
        else:
            chol.raise_if_failed()
        return out

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Cold figure (disclosed, never `value`): the first 20 passes after set-up, before the clock ramp.
    step()
    barrier()
    t_cold = time.perf_counter()
    for _ in range(20):
        step()
    barrier()
    cold_ms = (time.perf_counter() - t_cold) / 20 * 1e3

    # Clock ramp (set-up, untimed): the shader clock needs ~50 ms of load to reach its sustained value,
    # so the pipeline runs for 0.2 s before the W warm-up steps -- a short --steps run then measures the
    # same sustained rate as a long one.  Every rank runs the same number of passes (collectives match).
    t_ramp = time.perf_counter()
    n_ramp = 0
    while n_ramp < 2000:
        for _ in range(25):
            step()
        n_ramp += 25
        torch.cuda.synchronize()
        go_on = torch.tensor([1.0 if time.perf_counter() - t_ramp < 0.2 else 0.0], dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(go_on, op=dist.ReduceOp.MIN)
        if float(go_on.item()) == 0.0:
            break
    for _ in range(args.warmup):
        step()
    barrier()
    note("warm-up done (%d clock-ramp passes before the %d warm-up steps)" % (n_ramp, args.warmup))
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    check_info(args.steps - 1)
    barrier()
    dt_local = time.perf_counter() - t0
    cur[0] = None
    per_rank = [dt_local]
    seen = 1
    if world > 1:
        tl = torch.tensor([dt_local], dtype=torch.float64, device=coll_dev)
        allt = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(allt, tl)
        per_rank = [float(t.item()) for t in allt]
        if backend == "nccl":      # counted through an RCCL collective only (a gloo rehearsal reports null)
            ones = torch.ones(1, dtype=torch.float64, device=device)
            dist.all_reduce(ones)
            seen = int(round(float(ones.item())))
        else:
            seen = None
    dt = max(per_rank)          # MAX over ranks

    ms_hess = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_chol = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    ms_trsm = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))
    ms_ar = float(np.mean([e[4].elapsed_time(e[5]) for e in ev])) if world > 1 else 0.0

    leg3 = leg4 = None
    if world > 1:
        if not args.no_c3:
            note("configs[2] leg")
            leg3 = config3_leg(torch, dist, va, DeviceCholesky, world, rank, device, group, coll_dev,
                               args.c3_rows or 10_000_000, args.c3_dim)
        if not args.no_c4:
            note("configs[3] leg")
            leg4 = config4_leg(torch, dist, va, world, rank, device, group, coll_dev, args.c4_rows, args.hvp_dim)
    if rank == 0:
        traffic, traffic_source = latest_traffic(N, D)
        ms_per_step = dt / args.steps * 1e3
        value = world * N / (dt / args.steps)
        # dominant kernel: the MFMA TRSM (algorithmic flops 2 D^2 N, bytes 16 N D + 8 N)
        trsm_flops = 2.0 * D * D * N
        trsm_tflops = trsm_flops / (ms_trsm * 1e-3) / 1e12
        syrk_flops = float(N) * D * (D + 1)
        ms_syrk = ms_hess - ms_ar          # the fused-pass stage minus the collective inside it
        result = {
            "metric": "IJ sensitivities/sec (all-N dtheta/dw) at N=1M,D=100",
            "value": value, "unit": "sensitivities/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "weighted logistic regression N=%d D=%d fp64 per GPU "
                                   "(Hessian SYRK + multi-RHS TRSM), ridge prior 1" % (N, D),
                       "rows_per_gpu": N, "dim": D, "parallelism": "row-sharded dp%d" % world,
                       "backend": backend if world > 1 else None,
                       "collective_backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                       "grad_norm_at_theta_hat": grad_norm},
            "clock_ramp_steps": n_ramp + 21, "cold_ms_per_step": cold_ms,
            "rccl_ranks_seen": seen, "ranks": world, "per_rank_ms_per_step": [t / args.steps * 1e3 for t in per_rank],
            "roofline": {"bound": "mfma", "kernel": "k_trsm_rows_small (K2+K5 sens TRSM)",
                         "achieved": trsm_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": trsm_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes": 16.0 * N * D + 8.0 * N, "algorithmic_flops": trsm_flops,
                         "ms": ms_trsm,
                         "hbm_GBs": (16.0 * N * D + 8.0 * N) / (ms_trsm * 1e-3) / 1e9},
            "stages_ms": {"fused_link_grad_hessian": ms_syrk, "all_reduce_H_g": ms_ar,
                          "cholesky_pack": ms_chol, "sens_trsm": ms_trsm},
            "syrk": {"TFLOPs_triangle": syrk_flops / (ms_syrk * 1e-3) / 1e12,
                     "TFLOPs_full_square": 2.0 * N * D * D / (ms_syrk * 1e-3) / 1e12,
                     "frac_of_peak_triangle": syrk_flops / (ms_syrk * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                     "hbm_GBs": (8.0 * N * D + 24.0 * N) / (ms_syrk * 1e-3) / 1e9},
        }
        # spot check of the generator's contract: 1000 rows of this rank regenerated on the host (NumPy Philox)
        ridx = np.linspace(0, N - 1, 1000).astype(np.int64)
        Xr, yr = va.synth.glm_rows_host(world * N, D, "logistic", 0, ridx + rank * N)
        result["synthetic"] = {
            "generator": "Philox-4x32-10 counter-based (vt_synth_normal / vt_synth_response), seed 0, "
                         "X ~ N(0,1), theta* ~ N(0,1/D), y ~ Bernoulli(sigmoid(x.theta*)), w = 1, ridge prior 1",
            "host_regenerated_rows": int(len(ridx)),
            "X_max_abs_diff_vs_host": float(np.abs(X[torch.as_tensor(ridx, device=device)].cpu().numpy() - Xr).max()),
            "y_mismatches_vs_host": int((y[torch.as_tensor(ridx, device=device)].cpu().numpy() != yr).sum())}
        if world == 1:
            # what this box streams with torch's own kernels on arrays of the solve's size (context for `hbm_GBs`: the
            # dominant kernel reads X and writes S, a device copy is the same traffic with no arithmetic)
            def _rate(fn, nbytes, reps=20):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a0.record()
                for _ in range(reps):
                    fn()
                a1.record()
                torch.cuda.synchronize()
                return nbytes / (a0.elapsed_time(a1) / reps * 1e-3) / 1e9
            _tmp = torch.empty_like(S)
            result["roofline"]["stream_reference"] = {
                "device_copy_GBs_read_plus_write": _rate(lambda: _tmp.copy_(X), 16.0 * N * D),
                "read_only_sum_GBs": _rate(lambda: X.sum(), 8.0 * N * D),
                "note": "torch copy / sum of the same N x D arrays on this box, measured after the timed region: the "
                        "solve's read + write stream (hbm_GBs) against what a plain copy reaches"}
            del _tmp
        if world == 1:   # side measurements and CPU legs run at N=1 only: no rank waits on rank 0
            if not args.no_cpu_baseline:
                n_s = min(args.cpu_rows, N)
                Xh = X[:n_s].cpu().numpy()
                yh = y[:n_s].cpu().numpy()
                wh = w[:n_s].cpu().numpy()
                theta_h = theta.cpu().numpy()
                base, H_cpu, S_cpu = cpu_baseline(Xh, yh, wh, theta_h, prior, n_s)
                result["cpu_baseline"] = base
                if n_s == N:
                    # same rows on both sides -> full parity check of H and S
                    Hg = out["H"].cpu().numpy()
                    Sg = S.cpu().numpy()
                    result["parity"] = {
                        "H_max_rel_err": float(np.abs(Hg - H_cpu).max() / np.abs(H_cpu).max()),
                        "S_max_rel_err": float(np.abs(Sg.T - S_cpu).max() / np.abs(S_cpu).max()),
                        "rows_checked": int(N), "tolerance": 1e-10}
                    del Sg
                del Xh, H_cpu, S_cpu
                result["speedup_vs_cpu_baseline"] = value / base["value"]
                result["config1"] = config1_side_measurement(torch)
            if not args.no_fp32 and D <= 128:
                result["fp32_mode"] = fp32_side_measurement(torch, va, DeviceCholesky, X, y, w, theta, prior,
                                                            out["H"], S)
            del S, X
            torch.cuda.empty_cache()
            if not args.no_c3:
                result["config3_share"] = config3_leg(torch, dist, va, DeviceCholesky, 1, 0, device, None, coll_dev,
                                                      args.c3_rows or 1_250_000, args.c3_dim)
            if not args.no_hvp:
                result["hvp"], result["cg"] = hvp_cg_side_measurement(torch, device, args.hvp_rows, args.hvp_dim,
                                                                      args.cg_cpu_rows)
                torch.cuda.empty_cache()
            if not args.no_block:
                result["block"] = block_side_measurement(torch, args.block_groups)
        if leg3 is not None:
            result["config3_strong"] = leg3
        if leg4 is not None:
            result["config4_strong"] = leg4
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
