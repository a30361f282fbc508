"""Prototype timing: Hessian of a 256 < D <= 384 design as (one-pass kernel on the first 256 columns, the rest of theta
folded into the offset) + (strip X2^T diag(d) X by vt_gemm_tn), against the tiled route."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd import rows
N = 2_000_000
def timeit(fn, reps=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for D in [int(a) for a in sys.argv[1:]] or [257, 288, 320, 352]:
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    def full():
        obj._cache_key = None
        return obj.fused(theta, w)
    t_full = timeit(full)
    Href = full()["H"].clone()
    X1, X2 = obj.X[:, :256], obj.X[:, 256:D]
    t_off = timeit(lambda: rows.gemv(X2, theta[256:].contiguous()))
    off = rows.gemv(X2, theta[256:].contiguous())
    obj1 = va.WeightedGLMObjective(X1, y, family="logistic", prior=1.0, offset=off)
    def part1():
        obj1._cache_key = None
        return obj1.fused(theta[:256].contiguous(), w)
    t_1 = timeit(part1)
    out1 = part1()
    C = torch.empty((D - 256, D), dtype=torch.float64, device="cuda")
    t_2 = timeit(lambda: rows.gemm_tn(X2, obj.X[:, :D], rowscale=out1["d"], out=C))
    rows.gemm_tn(X2, obj.X[:, :D], rowscale=out1["d"], out=C)
    e1 = float((out1["H"] - Href[:256, :256]).abs().max() / Href.abs().max())
    Cref = Href[256:, :].clone(); Cref[:, 256:] -= torch.eye(D - 256, dtype=torch.float64, device="cuda")
    e2 = float((C - Cref).abs().max() / Href.abs().max())
    print("D=%d tiled %.3f ms | offset gemv %.3f + one-pass(256) %.3f + strip %.3f = %.3f ms | err %.1e %.1e"
          % (D, t_full, t_off, t_1, t_2, t_off + t_1 + t_2, e1, e2), flush=True)
    del X, obj, obj1, C, Href, out1
    torch.cuda.empty_cache()
