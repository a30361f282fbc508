import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd.linalg import DeviceCholesky
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, D = int(sys.argv[1]), int(sys.argv[2])
X = torch.randn((N, D), dtype=torch.float64, device="cuda")
y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
w = torch.ones(N, dtype=torch.float64, device="cuda")
theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
def f():
    obj._cache_key = None
    return obj.fused(theta, w)
ms = timeit(f); out = f()
print("N=%d D=%d fused link+grad+SYRK: %.2f ms (%.1f TF triangle)" % (N, D, ms, N*D*(D+1)/ms/1e9))
H = out["H"]
ms = timeit(lambda: DeviceCholesky.factor(H, check=False, pack=True))
print("potrf+pack: %.2f ms" % ms)
chol = DeviceCholesky.factor(H, pack=True)
S = chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0)      # the library's own (16-byte pitched) layout
ms = timeit(lambda: chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0, out=S))
print("trsm_rows: %.2f ms (%.1f TF)" % (ms, 2.0*N*D*D/ms/1e9))
