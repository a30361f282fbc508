"""Dev tool: whole-call time of the fp32-mode fused Hessian pass (and its TRSM) at N = 10^6.
python tools/time_hess32.py [D]   (VITTLES_HIP_LIB selects a dev build)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd.linalg import DeviceCholesky
N, D = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 100
X = torch.randn((N, D), dtype=torch.float32, device="cuda")
y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
w = torch.ones(N, dtype=torch.float64, device="cuda")
theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
def timeit(f, reps=200):
    for _ in range(100): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
tag = os.environ.get("VITTLES_HIP_LIB", "default")[-22:]
for fam in ("gaussian", "logistic"):
    obj = va.WeightedGLMObjective(X, y, family=fam, prior=1.0, dtype=torch.float32)
    def f():
        obj._cache_key = None
        return obj.fused(theta, w)
    print(tag, "D=%d" % D, fam, "hessian_f32 %.3f ms" % timeit(f), flush=True)
out = f()
chol = DeviceCholesky.factor(out["H"], pack=True)
S = torch.empty_like(X)
print(tag, "D=%d" % D, "trsm_f32 %.3f ms" % timeit(lambda: chol.solve_rows(X, rowscale=out["r"], alpha=-1.0, out=S)), flush=True)
