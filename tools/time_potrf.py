"""Dev tool: potrf + pack of the D x D Hessian alone (VITTLES_HIP_LIB selects an ablated build; results are then wrong)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vittles_amd.linalg import DeviceCholesky
tag = os.environ.get("VITTLES_HIP_LIB", "default")[-12:]
for D in [int(a) for a in sys.argv[1:]] or [100]:
    A = torch.randn((D, D), dtype=torch.float64, device="cuda")
    H = A @ A.T + D * torch.eye(D, dtype=torch.float64, device="cuda")
    f = lambda: DeviceCholesky.factor(H, check=False, pack=True)
    for _ in range(50): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500): f()
    e1.record(); torch.cuda.synchronize()
    print(tag, "D=%d potrf+pack %.2f us" % (D, e0.elapsed_time(e1) / 500 * 1e3), flush=True)
