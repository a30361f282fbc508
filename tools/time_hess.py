import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
N, D = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 100
X = torch.randn((N, D), dtype=torch.float64, device="cuda")
y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
w = torch.ones(N, dtype=torch.float64, device="cuda")
theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
for fam in ("gaussian", "logistic"):
    obj = va.WeightedGLMObjective(X, y, family=fam, prior=1.0)
    def f():
        obj._cache_key = None
        return obj.fused(theta, w)
    for _ in range(100): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    print(os.environ.get("VITTLES_HIP_LIB", "default")[-22:], fam, "%.3f ms" % (e0.elapsed_time(e1) / 200))
