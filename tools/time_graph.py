"""Dev tool: the headline step (fused Hessian -> potrf + pack -> solve) as eager launches against a captured
HIP graph of the same three calls (N = 10^6, D = 100)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd.linalg import DeviceCholesky
N, D = 1_000_000, 100
X, y, theta_star = va.synth.glm_problem(N, D, "logistic", seed=1, device="cuda")
w = torch.ones(N, dtype=torch.float64, device="cuda")
theta = 0.5 * theta_star
obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
S = torch.empty((N, D), dtype=torch.float64, device="cuda")
def step():
    obj.invalidate_cache()
    out = obj.fused(theta, w)
    chol = DeviceCholesky.factor(out["H"], check=False, pack=True)
    chol.solve_rows(obj.X, rowscale=out["r"], alpha=-1.0, out=S)
def timeit(fn, reps=300):
    for _ in range(300): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("eager  %.4f ms/step" % timeit(step), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(5): step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    step()
torch.cuda.synchronize()
ref = S.clone()
g.replay(); torch.cuda.synchronize()
print("graph replay equals eager:", bool((S == ref).all()))
print("graph  %.4f ms/step" % timeit(g.replay), flush=True)
print("eager  %.4f ms/step" % timeit(step), flush=True)
