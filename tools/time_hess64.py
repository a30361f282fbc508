"""Dev tool: whole-call time of the fp64 fused Hessian pass at N = 10^6 (200 repetitions after 50 warm-up calls).
python tools/time_hess64.py [D ...]   (VITTLES_HIP_LIB selects a dev build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
N = 1_000_000
Ds = [int(a) for a in sys.argv[1:]] or [100]
tag = os.environ.get("VITTLES_HIP_LIB", "default")[-14:]
for D in Ds:
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
    for fam in ("logistic", "gaussian"):
        obj = va.WeightedGLMObjective(X, y, family=fam, prior=1.0)
        def f():
            obj._cache_key = None
            return obj.fused(theta, w)
        for _ in range(50): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        print(tag, "D=%d %s hessian %.4f ms (%.1f TF tri)" % (D, fam, ms, N * D * (D + 1) / ms / 1e9), flush=True)
