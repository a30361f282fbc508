"""Dev check of one Hessian build against plain fp64 torch (usage: VITTLES_HIP_LIB=... python tools/check_hess.py D [N])"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
D = int(sys.argv[1]) if len(sys.argv) > 1 else 100
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100003
torch.manual_seed(D)
X = torch.randn((N, D), dtype=torch.float64, device="cuda")
y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
w = torch.rand(N, dtype=torch.float64, device="cuda") + 0.5
theta = torch.randn(D, dtype=torch.float64, device="cuda") / D ** 0.5
obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
f = obj.fused(theta, w)
z = X @ theta
p = torch.sigmoid(z)
d = w * p * (1 - p)
H = X.t() @ (d[:, None] * X) + torch.eye(D, dtype=torch.float64, device="cuda")
g = X.t() @ (w * (p - y)) + theta
eh = float((f["H"] - H).abs().max() / H.abs().max())
eg = float((f["g"] - g).abs().max() / g.abs().max())
ed = float((f["d"] - d).abs().max())
print(os.environ.get("VITTLES_HIP_LIB", "default")[-14:], "D=%d N=%d H %.1e g %.1e d %.1e" % (D, N, eh, eg, ed), "OK" if max(eh, eg, ed) < 1e-12 else "FAIL")
sys.exit(0 if max(eh, eg, ed) < 1e-12 else 1)
