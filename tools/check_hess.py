"""Dev tool: parity of the fused Hessian pass vs the oracle for a list of D (N large enough for the tile kernel).
python tools/check_hess.py 100 97 104 112"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vittles_amd as va
from oracle import glm

Ds = [int(a) for a in sys.argv[1:]] or [100]
worst = 0.0
for D in Ds:
    for fam, N, off in (("logistic", 20011, False), ("poisson", 4100, True), ("gaussian", 9999, False)):
        X, y, w, _ = glm.synth(N, D, fam, seed=D + N)
        rng = np.random.default_rng(D)
        w = rng.uniform(0.2, 1.8, N)
        theta = rng.standard_normal(D) / np.sqrt(D) * (0.3 if fam == "poisson" else 1.0)
        offset = rng.standard_normal(N) * 0.1 if off else None
        obj = va.WeightedGLMObjective(X, y, family=fam, prior=0.7, scale=1.3, offset=offset)
        out = obj.fused(theta, w)
        H = glm.hessian(fam, X, y, theta, w, 0.7, 1.3, offset)
        g = glm.grad(fam, X, y, theta, w, 0.7, 1.3, offset)
        _, l1, l2 = glm.link(fam, glm.linear_predictor(X, theta, offset), y)
        errs = [np.abs(out["H"].cpu().numpy() - H).max() / np.abs(H).max(),
                np.abs(out["g"].cpu().numpy() - g).max() / np.abs(g).max(),
                np.abs(out["r"].cpu().numpy() - 1.3 * l1).max() / np.abs(l1).max(),
                np.abs(out["d"].cpu().numpy() - 1.3 * w * l2).max() / np.abs(w * l2).max()]
        worst = max(worst, max(errs))
        print("D=%d %s N=%d: H %.1e g %.1e r %.1e d %.1e" % (D, fam, N, *errs), flush=True)
print("worst", worst, "OK" if worst < 1e-10 else "FAIL")
sys.exit(0 if worst < 1e-10 else 1)
