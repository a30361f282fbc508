"""Dev tool: the MFMA-resident sensitivity solve alone (N = 4*10^6 by default): python tools/time_trsm_small.py [N] [D ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vittles_amd.linalg import DeviceCholesky
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
Ds = [int(a) for a in sys.argv[2:]] or [20]
tag = os.environ.get("VT_TRSM_WGS", "-")
for D in Ds:
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    r = torch.randn(N, dtype=torch.float64, device="cuda")
    A = torch.randn((D, D), dtype=torch.float64, device="cuda")
    H = A @ A.T + D * torch.eye(D, dtype=torch.float64, device="cuda")
    chol = DeviceCholesky.factor(H, pack=True)
    S = torch.empty_like(X)
    f = lambda: chol.solve_rows(X, rowscale=r, alpha=-1.0, out=S)
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 100
    print("wgs=%s D=%d trsm_rows %.4f ms (%.1f TF, %.0f GB/s)" % (tag, D, ms, 2.0 * N * D * D / ms / 1e9, 16.0 * N * D / ms / 1e6), flush=True)
    del X, S
