"""Time individual stages on the GPU (dev tool): python tools/time_stages.py [N] [D ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd.linalg import DeviceCholesky

def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Ds = [int(a) for a in sys.argv[2:]] or [100]
for D in Ds:
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
    for fam in ("logistic", "gaussian", "poisson"):
        obj = va.WeightedGLMObjective(X, y, family=fam, prior=1.0)
        def f():
            obj._cache_key = None
            return obj.fused(theta, w)
        ms = timeit(f)
        print("D=%d %s fused hessian: %.3f ms  (%.2f TF tri, %.0f GB/s)" % (D, fam, ms, N*D*(D+1)/ms/1e9, 8.0*N*D/ms/1e6))
    out = f()
    H = obj.fused(theta, w)["H"]
    ms = timeit(lambda: DeviceCholesky.factor(H, check=False))
    print("D=%d potrf+pack fused %.3f ms" % (D, timeit(lambda: DeviceCholesky.factor(H, check=False, pack=True))))
    chol = DeviceCholesky.factor(H)
    def p():
        chol._pack = None
        chol.pack()
    print("D=%d potrf %.3f ms, pack %.3f ms" % (D, ms, timeit(p)))
    S = torch.empty((N, D), dtype=torch.float64, device="cuda")
    r = out["r"]
    ms = timeit(lambda: chol.solve_rows(X, rowscale=r, alpha=-1.0, out=S))
    print("D=%d trsm_rows: %.3f ms (%.2f TF alg, %.0f GB/s)" % (D, ms, 2.0*N*D*D/ms/1e9, 16.0*N*D/ms/1e6))
    del X, S
