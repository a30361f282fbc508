"""Time the block-sparse VI path at BASELINE.json configs[4] size (dev tool):
python tools/time_block.py [G] [ng] [k]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vittles_amd as va
from oracle import vi

G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 16
k = int(sys.argv[3]) if len(sys.argv) > 3 else 4
X, y = vi.synth(G, ng, k, 0)
m = vi.HierGaussMFVI(X, y, G, ng, k, 2.0, 1.5, 0.5)
obj = va.HierGaussMFVIObjective(X, y, G, ng, k, 2.0, 1.5, 0.5)
eta = m.optimum()
sb = va.SparseBlockHessian(obj, obj.sparsity_array())
eta_d = torch.tensor(eta, device="cuda")


def wall(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

ms, _ = wall(lambda: obj.hvp(eta_d, eta_d))
print("G=%d ng=%d k=%d d=%d: one HVP %.3f ms" % (G, ng, k, m.d, ms))
ms, parts = wall(lambda: sb._block_parts(eta_d))
print("device block assembly (%d coloured HVPs + gathers): %.3f ms" % (2 * k, ms))
ms, gparts = wall(lambda: sb._global_parts(eta_d, obj.global_inds()))
print("device global rows (%d HVPs + gathers): %.3f ms" % (2 * k, ms))
ms, H = wall(lambda: sb.get_hessian(eta), reps=2)
print("get_hessian incl. scipy COO construction and block+global sum on the host: %.1f ms (nnz %d)" % (ms, H.nnz))
t0 = time.perf_counter()
lr = va.LinearResponseCovariances(obj, eta, validate_optimum=True, hessian_at_opt=H, grad_tol=1e-6)
torch.cuda.synchronize()
print("LinearResponseCovariances set-up (arrowhead factor + Newton-step validation): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
# moments = the variational means of the first 8 groups and of mu
K = 8 * k + k
J = np.zeros((K, m.d))
for g in range(8):
    for j in range(k):
        J[g * k + j, 2 * k * g + j] = 1.0
for j in range(k):
    J[8 * k + j, 2 * k * G + j] = 1.0
ms, cov = wall(lambda: lr.get_lr_covariance_from_jacobians(J, J), reps=3)
print("LR covariance of %d moments: %.2f ms" % (K, ms))
