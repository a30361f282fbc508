import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vittles_amd.linalg import DeviceCholesky
D = int(sys.argv[1]) if len(sys.argv) > 1 else 100
A = torch.randn((4 * D, D), dtype=torch.float64, device="cuda")
H = A.T @ A + torch.eye(D, dtype=torch.float64, device="cuda")
for pack in (False, True):
    f = lambda: DeviceCholesky.factor(H, check=False, pack=pack)
    for _ in range(200): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): f()
    e1.record(); torch.cuda.synchronize()
    print("D=%d pack=%s: %.1f us" % (D, pack, e0.elapsed_time(e1) / 500 * 1e3))
