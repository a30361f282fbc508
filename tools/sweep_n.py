"""Dev tool: the dense pipeline (fused Hessian, potrf + pack, sensitivity solve) over a sweep of N at fixed D.
python tools/sweep_n.py [D] [N ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd.linalg import DeviceCholesky
D = int(sys.argv[1]) if len(sys.argv) > 1 else 100
Ns = [int(a) for a in sys.argv[2:]] or [1000, 4000, 4096, 16000, 64000, 256000, 1000000, 4000000]
def timeit(f, reps=50):
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for N in Ns:
    X, y, theta = va.synth.glm_problem(N, D, "logistic", seed=1)
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    def f():
        obj._cache_key = None
        return obj.fused(theta, w)
    th = timeit(f)
    out = f()
    tp = timeit(lambda: DeviceCholesky.factor(out["H"], check=False, pack=True))
    chol = DeviceCholesky.factor(out["H"], pack=True)
    S = chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0)
    tt = timeit(lambda: chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0, out=S))
    tot = th + tp + tt
    print("D=%d N=%8d hessian %.4f  potrf+pack %.4f  trsm %.4f  total %.4f ms -> %.3g sens/s" % (D, N, th, tp, tt, tot, N / tot * 1e3), flush=True)
    del X, obj, S
