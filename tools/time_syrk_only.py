"""Fused link+grad+Hessian only: usage python tools/time_syrk_only.py N D [D ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = int(sys.argv[1])
for D in map(int, sys.argv[2:]):
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    def f():
        obj._cache_key = None
        return obj.fused(theta, w)
    ms = timeit(f)
    print("N=%d D=%4d hess %7.3f ms %5.1f TF (%4.1f%%) C0=%s" % (N, D, ms, N*D*(D+1)/ms/1e9, N*D*(D+1)/ms/1e9/0.786, os.environ.get("VT_SYRK_C0")), flush=True)
    del X, obj
    torch.cuda.empty_cache()
