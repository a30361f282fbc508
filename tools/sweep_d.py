"""Dev tool: fused Hessian and sensitivity solve over a sweep of D at fixed N, to spot dispatch cliffs.
python tools/sweep_d.py [N] [D ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd.linalg import DeviceCholesky
PLAIN = os.environ.get("VT_SWEEP_PLAIN") is not None      # keep odd leading dimensions (no re-layout, dense S)
if PLAIN:
    va.objectives.RELAYOUT_UNALIGNED_ROWS = False
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
Ds = [int(a) for a in sys.argv[2:]] or list(range(1, 133, 3))
def timeit(f, reps=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for D in Ds:
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    def f():
        obj._cache_key = None
        return obj.fused(theta, w)
    th = timeit(f)
    out = f()
    chol = DeviceCholesky.factor(out["H"], pack=True)
    S = torch.empty((N, D), dtype=torch.float64, device="cuda") if PLAIN else chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0)
    tt = timeit(lambda: chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0, out=S))
    print("D=%3d hessian %.4f ms (%5.0f GB/s, %5.1f TF tri)   trsm %.4f ms (%5.0f GB/s, %5.1f TF)" % (
        D, th, 8.0 * N * D / th / 1e6, N * D * (D + 1) / th / 1e9, tt, 16.0 * N * D / tt / 1e6, 2.0 * N * D * D / tt / 1e9), flush=True)
    del X, S, obj, out, chol
