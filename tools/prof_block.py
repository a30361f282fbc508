import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import vittles_amd as va
from vittles_amd import sparse_hessian_lib as shl
from oracle import vi
from scipy.sparse import coo_matrix
G, ng, k = 10000, 16, 4
X, y = vi.synth(G, ng, k, 0)
m = vi.HierGaussMFVI(X, y, G, ng, k, 2.0, 1.5, 0.5)
obj = va.HierGaussMFVIObjective(X, y, G, ng, k, 2.0, 1.5, 0.5)
eta = m.optimum()
sb = va.SparseBlockHessian(obj, obj.sparsity_array())
for _ in range(3): H = sb.get_hessian(eta)
def T(f, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
print("get_hessian %.2f ms" % T(lambda: sb.get_hessian(eta))[0])
opt = torch.tensor(eta, device="cuda")
d = opt.shape[0]; M = 2 * k; Kg = 2 * k; nb = G * M * M; ngl = Kg * (2 * G * M + 2 * Kg)
vals = torch.empty(nb + ngl, dtype=torch.float64, device="cuda"); rows = torch.empty(nb + ngl, dtype=torch.int64, device="cuda"); cols = torch.empty_like(rows)
print("to_device(eta) %.2f" % T(lambda: torch.tensor(eta, device="cuda"))[0])
print("setdiff %.2f" % T(lambda: np.setdiff1d(np.arange(d), np.hstack(sb._sparsity_array)))[0])
print("block parts %.2f" % T(lambda: sb._block_parts(opt, 0, out=(vals[:nb], rows[:nb], cols[:nb])))[0])
gi = obj.global_inds()
print("global parts %.2f" % T(lambda: sb._global_parts(opt, gi, 0, out=(vals[nb:], rows[nb:], cols[nb:])))[0])
ms, trip = T(lambda: shl._triplets_to_host(vals, rows, cols, d)); print("triplets_to_host %.2f" % ms)
ms, Hc = T(lambda: coo_matrix(trip, (d, d))); print("coo_matrix %.2f" % ms)
print("fingerprint %.2f" % T(lambda: shl.matrix_fingerprint(Hc))[0])
