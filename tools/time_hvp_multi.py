"""Multi-vector HVP (vt_glm_hvp_multi) against K single products: usage python tools/time_hvp_multi.py N D K [K ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd import rows
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, D = int(sys.argv[1]), int(sys.argv[2])
X = torch.randn((N, D), dtype=torch.float64, device="cuda")
d = torch.rand(N, dtype=torch.float64, device="cuda")
v = torch.randn(D, dtype=torch.float64, device="cuda")
out1 = torch.empty(D, dtype=torch.float64, device="cuda")
ms1 = timeit(lambda: rows.glm_hvp(X, d, v, prior_scalar=1.0, out=out1))
print("N=%d D=%d single HVP %.3f ms (%.0f GB/s)" % (N, D, ms1, 8.0 * N * D / ms1 / 1e6), flush=True)
for K in map(int, sys.argv[3:]):
    V = torch.randn((K, D), dtype=torch.float64, device="cuda")
    out = torch.empty((K, D), dtype=torch.float64, device="cuda")
    ms = timeit(lambda: rows.glm_hvp_multi(X, d, V, prior_scalar=1.0, out=out))
    ref = rows.glm_hvp(X, d, V[K - 1].contiguous(), prior_scalar=1.0)
    err = float((out[K - 1] - ref).abs().max() / ref.abs().max())
    print("  K=%3d multi %.3f ms = %.2f single products (%.3f ms per vector, X read at %.0f GB/s per pass), last column vs single %.1e"
          % (K, ms, ms / ms1, ms / K, 16.0 * N * D / ms / 1e6, err), flush=True)
