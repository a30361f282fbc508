"""Time the second-order data-weight pass (vt_glm_taylor2_rhs) and the whole order-2 Taylor evaluation
(dev tool): python tools/time_taylor.py [NxD ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import vittles_amd as va
from vittles_amd import rows
for (N, D) in [(int(a.split('x')[0]), int(a.split('x')[1])) for a in (sys.argv[1:] or ["1000000x100", "1000000x1000"])]:
    X = torch.randn((N, D), dtype=torch.float64, device="cuda")
    y = (torch.rand(N, dtype=torch.float64, device="cuda") < 0.5).double()
    w = torch.ones(N, dtype=torch.float64, device="cuda")
    obj = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
    theta, _ = va.optim.newton_glm(obj, w)
    z, r, d = rows.glm_link(obj, theta, w, want_z=True)
    dw = torch.rand(N, dtype=torch.float64, device="cuda") - 0.5
    u = torch.randn(D, dtype=torch.float64, device="cuda")
    def timeit(f, reps=50):
        for _ in range(10): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = timeit(lambda: rows.glm_taylor2_rhs(obj, z, w, dw, u))
    print("taylor2 rhs N=%d D=%d: %.3f ms  %.0f GB/s (algorithmic 8ND+24N+16D)" % (N, D, ms, (8.0*N*D+24*N+16*D)/ms/1e6))
    te = va.ParametricSensitivityTaylorExpansion.optimization_objective(obj, theta, w, order=2)
    ms = timeit(lambda: te.evaluate_taylor_series(w + dw), reps=20)
    print("   evaluate_taylor_series (orders 1+2: two passes over X, two solves): %.3f ms" % ms)
    del X
