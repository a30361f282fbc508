"""Dev tool: the large-D blocked TRSM alone (random SPD H, no parity check).  python tools/time_trsm_large.py N D"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vittles_amd.linalg import DeviceCholesky
N, D = int(sys.argv[1]), int(sys.argv[2])
X = torch.randn((N, D), dtype=torch.float64, device="cuda")
A = torch.randn((2 * D, D), dtype=torch.float64, device="cuda")
H = A.t() @ A / D + torch.eye(D, dtype=torch.float64, device="cuda")
r = torch.rand(N, dtype=torch.float64, device="cuda")
chol = DeviceCholesky.factor(H, check=False, pack=True)
S = torch.empty((N, D), dtype=torch.float64, device="cuda")
f = lambda: chol.solve_rows(X, rowscale=r, alpha=-1.0, out=S)
f(); f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(os.environ.get("VITTLES_HIP_LIB", "default")[-20:], os.environ.get("VT_LARGE_TM128", "-"), "trsm_rows N=%d D=%d: %.2f ms (%.1f TF)" % (N, D, ms, 2.0 * N * D * D / ms / 1e9), flush=True)
