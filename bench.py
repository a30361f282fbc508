#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X linear-response sensitivity hot path.

Workload (BASELINE.json configs[1]): weighted logistic regression, N = 1,000,000 rows,
D = 100, fp64, ridge prior 1: one "step" = all-N infinitesimal-jackknife sensitivities
d theta-hat / d w  from device-resident (X, y, w, theta-hat) to device-resident S:
    fused link+gradient+Hessian SYRK  ->  (all-reduce of H when row-sharded)
    -> Cholesky -> TRSM operand pack -> MFMA multi-RHS TRSM  S = -H^{-1} (diag(r) X)^T
Metric: IJ sensitivities / second = (rows over all ranks) / (max-over-ranks step time).
With --gpus N > 1 every rank holds its own N-row shard (weak scaling); the only
collective is the RCCL all-reduce of the D x D partial Hessian.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
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
    p.add_argument("--no-hvp", action="store_true", help="skip the HVP GB/s side measurement")
    p.add_argument("--hvp-rows", type=int, default=100_000)
    p.add_argument("--hvp-dim", type=int, default=20_000)
    return p.parse_args()


def cpu_baseline(Xh, yh, wh, theta_h, prior, S_dev_rows, n_sample):
    """Time the oracle (CPU restatement: NumPy Hessian + the reference's LAPACK calls
    cho_factor/cho_solve with nrhs = N) on a bounded sample of the same workload and
    use it as the parity checker for the GPU result."""
    from oracle import glm
    import threadpoolctl
    n = min(n_sample, Xh.shape[0])
    Xs, ys, ws = Xh[:n], yh[:n], wh[:n]
    glm.sensitivity("logistic", Xs[:20000], ys[:20000], theta_h, ws[:20000], prior=prior)   # warm-up
    t0 = time.perf_counter()
    H, jac0, S = glm.sensitivity("logistic", Xs, ys, theta_h, ws, prior=prior)
    dt = time.perf_counter() - t0
    threads = 1
    try:
        infos = threadpoolctl.threadpool_info()
        threads = max([i.get("num_threads", 1) for i in infos if i.get("user_api") == "blas"] or [1])
    except Exception:
        pass
    return dict(value=n / dt, unit="sensitivities/s", cores=int(threads), kind="port",
                sample="oracle.glm.sensitivity (NumPy X^T D X + scipy cho_factor/cho_solve, nrhs=N) on the first "
                       "%d rows of the same N x D workload, %.2f s" % (n, dt)), H, S


def hvp_side_measurement(torch, device, n_rows, dim, reps=10):
    """The metric's second figure: fused HVP kernel (BASELINE.json configs[3], D = 20k
    ridge-logistic CG) in algorithmic GB/s = (8ND + 8N + 16D) / kernel time, HIP events on
    the launch stream.  Runs after the timed region of the headline step, rank 0 only."""
    from vittles_amd import rows
    X = torch.randn((n_rows, dim), dtype=torch.float64, device=device)
    d = torch.rand(n_rows, dtype=torch.float64, device=device)
    v = torch.randn(dim, dtype=torch.float64, device=device)
    out = torch.empty(dim, dtype=torch.float64, device=device)
    for _ in range(3):
        rows.glm_hvp(X, d, v, prior_scalar=1.0, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        rows.glm_hvp(X, d, v, prior_scalar=1.0, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = 8.0 * n_rows * dim + 8.0 * n_rows + 16.0 * dim
    gbs = nbytes / (ms * 1e-3) / 1e9
    del X
    return {"kernel": "k_rows_wide<OP_HVP> (K6 fused HVP)", "rows": n_rows, "dim": dim, "ms": ms,
            "algorithmic_bytes": nbytes, "GBs": gbs, "peak_GBs": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS}


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
        obj._cache_key = None
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
            "S_max_rel_err_gpu_vs_ad": err}


def note(msg):
    if os.environ.get("VITTLES_BENCH_VERBOSE"):
        print("[bench rank %s %.1fs] %s" % (os.environ.get("RANK", "0"), time.perf_counter(), msg),
              file=sys.stderr, flush=True)


def main():
    args = parse()
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
    dev_index = local_rank % torch.cuda.device_count()
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

    note("process group up")
    import vittles_amd as va
    from vittles_amd.linalg import DeviceCholesky
    from vittles_amd import _lib, device as dev

    N, D = args.rows, args.dim
    prior = 1.0
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    X = torch.randn((N, D), dtype=torch.float64, device=device, generator=gen)
    gth = torch.Generator(device=device)
    gth.manual_seed(99)   # same theta* on every rank
    theta_star = torch.randn(D, dtype=torch.float64, device=device, generator=gth) / D ** 0.5
    y = (torch.rand(N, dtype=torch.float64, device=device, generator=gen) < torch.sigmoid(X @ theta_star)).double()
    w = torch.ones(N, dtype=torch.float64, device=device)
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=prior, process_group=group)

    # theta-hat: damped Newton on the device with the same kernels (set-up, untimed).
    theta, _info = va.optim.newton_glm(obj, w, grad_tol=1e-10)
    grad_norm = float(obj.fused(theta.clone(), w)["g"].norm())
    note("theta-hat found, |g| = %.2e" % grad_norm)

    lib = _lib.load()
    S = torch.empty((N, D), dtype=torch.float64, device=device)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
           torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]

    # potrf's LAPACK-style info of every step is read on the host inside the timed region, one
    # step late (pinned buffer + event), so the check never drains the stream
    info_host = torch.zeros(max(args.steps, 1), dtype=torch.int32).pin_memory()
    info_ev = [None] * args.steps

    def check_info(i):
        info_ev[i].synchronize()
        DeviceCholesky.raise_for_info(int(info_host[i]))

    def step(i=None):
        obj._cache_key = None               # no cached pass: every step streams X
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

    # Clock ramp (set-up, untimed): the shader clock needs ~50 ms of load to reach its sustained value,
    # so the pipeline runs for 0.2 s before the W warm-up steps -- a short --steps run then measures the
    # same sustained rate as a long one.  Every rank runs the same number of passes (collectives match).
    barrier()
    t_ramp = time.perf_counter()
    n_ramp = 0
    while n_ramp < 2000:
        for _ in range(25):
            step()
        n_ramp += 25
        torch.cuda.synchronize()
        go_on = torch.tensor([1.0 if time.perf_counter() - t_ramp < 0.2 else 0.0], dtype=torch.float64,
                             device=device if backend == "nccl" else "cpu")
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
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    ms_hess = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))
    ms_chol = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))
    ms_trsm = float(np.mean([e[2].elapsed_time(e[3]) for e in ev]))

    if rank == 0:
        # HBM traffic of the dominant kernel: PMC counters can only be collected under rocprofv3
        # (separate --pmc passes), so the per-launch figure measured for this exact workload is
        # read from profiles/ and reported as null for any other shape.
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj["workload"] == {"rows": N, "dim": D}:
                traffic = tj["k_trsm_rows_small"]["hbm_bytes_per_launch"]
        except Exception:
            traffic = None
        ms_per_step = dt / args.steps * 1e3
        value = world * N / (dt / args.steps)
        # dominant kernel: the MFMA TRSM (algorithmic flops 2 D^2 N, bytes 16 N D + 8 N)
        trsm_flops = 2.0 * D * D * N
        trsm_tflops = trsm_flops / (ms_trsm * 1e-3) / 1e12
        syrk_flops = float(N) * D * (D + 1)
        result = {
            "metric": "IJ sensitivities/sec (all-N dtheta/dw) at N=1M,D=100",
            "value": value, "unit": "sensitivities/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "weighted logistic regression N=%d D=%d fp64 per GPU "
                                   "(Hessian SYRK + multi-RHS TRSM), ridge prior 1" % (N, D),
                       "rows_per_gpu": N, "dim": D, "parallelism": "row-sharded dp%d" % world,
                       "grad_norm_at_theta_hat": grad_norm},
            "roofline": {"bound": "mfma", "kernel": "k_trsm_rows_small (K2+K5 sens TRSM)",
                         "achieved": trsm_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": trsm_tflops / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "algorithmic_bytes": 16.0 * N * D + 8.0 * N, "algorithmic_flops": trsm_flops,
                         "ms": ms_trsm,
                         "hbm_GBs": (16.0 * N * D + 8.0 * N) / (ms_trsm * 1e-3) / 1e9},
            "stages_ms": {"fused_link_grad_hessian": ms_hess, "cholesky_pack": ms_chol, "sens_trsm": ms_trsm},
            "syrk": {"TFLOPs_triangle": syrk_flops / (ms_hess * 1e-3) / 1e12,
                     "TFLOPs_full_square": 2.0 * N * D * D / (ms_hess * 1e-3) / 1e12,
                     "hbm_GBs": (8.0 * N * D + 24.0 * N) / (ms_hess * 1e-3) / 1e9},
        }
        if not args.no_hvp and world == 1:   # side measurements run at N=1 only: no rank waits on rank 0
            result["hvp"] = hvp_side_measurement(torch, device, args.hvp_rows, args.hvp_dim)
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed on rank 0 at N=1 only
            n_s = min(args.cpu_rows, N)
            Xh = X[:n_s].cpu().numpy()
            yh = y[:n_s].cpu().numpy()
            wh = w[:n_s].cpu().numpy()
            theta_h = theta.cpu().numpy()
            base, H_cpu, S_cpu = cpu_baseline(Xh, yh, wh, theta_h, prior, None, n_s)
            result["cpu_baseline"] = base
            if n_s == N and world == 1:
                # same rows on both sides -> full parity check of H and S
                Hg = out["H"].cpu().numpy()
                Sg = S.cpu().numpy()
                result["parity"] = {
                    "H_max_rel_err": float(np.abs(Hg - H_cpu).max() / np.abs(H_cpu).max()),
                    "S_max_rel_err": float(np.abs(Sg.T - S_cpu).max() / np.abs(S_cpu).max()),
                    "rows_checked": int(N), "tolerance": 1e-10}
            result["speedup_vs_cpu_baseline"] = value / base["value"]
            result["config1"] = config1_side_measurement(torch)
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
