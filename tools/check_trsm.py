"""Dev check + timing of the all-rows solve at one D against a NumPy solve (usage: python tools/check_trsm.py D [N])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vittles_amd.linalg import DeviceCholesky
D = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
torch.manual_seed(D)
A = torch.randn((2 * D, D), dtype=torch.float64, device="cuda")
H = A.T @ A / D + torch.eye(D, dtype=torch.float64, device="cuda")
X = torch.randn((N, D), dtype=torch.float64, device="cuda")
r = torch.rand(N, dtype=torch.float64, device="cuda") + 0.5
chol = DeviceCholesky.factor(H, pack=True)
S = chol.solve_rows(X, rowscale=r, alpha=-1.0)
idx = torch.tensor([0, 1, 17, N // 2, N - 2, N - 1], device="cuda")
ref = -(r[idx, None] * torch.linalg.solve(H, X[idx].T).T)
err = float((S[idx] - ref).abs().max() / ref.abs().max())
for _ in range(5): chol.solve_rows(X, rowscale=r, alpha=-1.0, out=S)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): chol.solve_rows(X, rowscale=r, alpha=-1.0, out=S)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("%s D=%d N=%d solve %.3f ms %.1f TF (%.1f%%) err %.1e %s" % (os.environ.get("VITTLES_HIP_LIB", "default")[-12:], D, N, ms, 2.0 * N * D * D / ms / 1e9, 2.0 * N * D * D / ms / 1e9 / 0.786, err, "OK" if err < 1e-11 else "FAIL"))
