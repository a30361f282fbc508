import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vittles_amd import rows
for (N, D) in [(int(a.split('x')[0]), int(a.split('x')[1])) for a in (sys.argv[1:] or ["100000x20000", "1000000x1000", "4000000x100"])]:
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    d = torch.rand(N, dtype=torch.float64, device="cuda")
    v = torch.randn(D, dtype=torch.float64, device="cuda")
    out = torch.empty(D, dtype=torch.float64, device="cuda")
    f = lambda: rows.glm_hvp(X, d, v, prior_scalar=1.0, out=out)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("HVP N=%d D=%d: %.3f ms  %.0f GB/s (algorithmic 8ND+8N+16D)" % (N, D, ms, (8.0*N*D+8*N+16*D)/ms/1e6))
    del X
