"""fp32 storage mode, wide rows: fused HVP on a float X (usage: python tools/time_hvp32.py [N D])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vittles_amd import rows
N, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 20000)
X = torch.randn((N, D), dtype=torch.float64, device="cuda")
X32 = X.to(torch.float32)
d = torch.rand(N, dtype=torch.float64, device="cuda")
v = torch.randn(D, dtype=torch.float64, device="cuda")
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
o64 = torch.empty(D, dtype=torch.float64, device="cuda"); o32 = torch.empty_like(o64)
m64 = timeit(lambda: rows.glm_hvp(X, d, v, prior_scalar=1.0, out=o64))
m32 = timeit(lambda: rows.glm_hvp(X32, d, v, prior_scalar=1.0, out=o32))
ref = rows.glm_hvp(X32.double(), d, v, prior_scalar=1.0)
print("N=%d D=%d HVP fp64 X %.3f ms (%.0f GB/s) | float X %.3f ms (%.0f GB/s), vs fp64 arithmetic on the rounded X %.1e, vs fp64 X %.1e"
      % (N, D, m64, 8.0 * N * D / m64 / 1e6, m32, 4.0 * N * D / m32 / 1e6,
         float((o32 - ref).abs().max() / ref.abs().max()), float((o32 - o64).abs().max() / o64.abs().max())))
