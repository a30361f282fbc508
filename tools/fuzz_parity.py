"""Randomised parity sweep of the dense pipeline against the oracle (dev tool, GPU):
python tools/fuzz_parity.py [cases] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vittles_amd as va
from oracle import glm

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
worst = {}
for it in range(cases):
    D = int(rng.choice([rng.integers(1, 129), rng.integers(1, 129), rng.integers(129, 257), rng.integers(257, 600)]))
    N = int(rng.choice([rng.integers(1, 200), rng.integers(200, 6000), rng.integers(4096, 30000)]))
    if D > 256:
        N = min(N, 6000)
    fam = str(rng.choice(["logistic", "poisson", "gaussian"]))
    X, y, w, _ = glm.synth(N, D, fam, seed=int(rng.integers(1 << 30)))
    w = 0.25 + rng.random(N)
    theta = rng.standard_normal(D) / np.sqrt(D) * (0.5 if fam == "poisson" else 1.0)
    offset = 0.1 * rng.standard_normal(N) if rng.random() < 0.4 else None
    prior = np.linspace(0.5, 2.0, D) if rng.random() < 0.7 else 1.0
    scale = float(rng.choice([1.0, 0.5, 2.5]))
    va.objectives.RELAYOUT_UNALIGNED_ROWS = bool(rng.random() < 0.5)   # half the cases keep odd / unaligned rows as they are
    strided = rng.random() < 0.25            # X as a column slice of a wider matrix: odd / unaligned rows
    if strided:
        big = torch.tensor(np.hstack([rng.standard_normal((N, 3)), X, rng.standard_normal((N, 2))]), device="cuda")
        Xd = big[:, 3:3 + D]
    else:
        Xd = torch.tensor(X, device="cuda")
    obj = va.WeightedGLMObjective(Xd, y, family=fam, prior=prior, scale=scale, offset=offset)
    out = obj.fused(theta, w)
    H_ref = glm.hessian(fam, X, y, theta, w, prior, scale, offset)
    g_ref = glm.grad(fam, X, y, theta, w, prior, scale, offset)
    la = va.HyperparameterSensitivityLinearApproximation(obj, theta, w)
    _, _, S_ref = glm.sensitivity(fam, X, y, theta, w, prior, scale, offset)
    errs = dict(H=rel(out["H"].cpu().numpy(), H_ref), g=rel(out["g"].cpu().numpy(), g_ref),
                S=rel(la.get_dopt_dhyper(), S_ref), val=abs(obj(theta, w) - glm.value(fam, X, y, theta, w, prior, scale, offset))
                / abs(glm.value(fam, X, y, theta, w, prior, scale, offset)))
    v = rng.standard_normal(D)
    from vittles_amd.cg import GLMHessianVectorProduct
    errs["hvp"] = rel(GLMHessianVectorProduct(obj, theta, w)(v), glm.hvp(fam, X, y, theta, w, v, prior, scale, offset))
    if True:           # fp32 storage mode of the same case: 1e-4 against the fp64 oracle (scaled by the conditioning)
        o32 = va.WeightedGLMObjective(Xd.to(torch.float32), y, family=fam, prior=prior, scale=scale, offset=offset,
                                      dtype=torch.float32)
        f32 = o32.fused(theta, w)
        la32 = va.HyperparameterSensitivityLinearApproximation(o32, theta, w)
        e32 = dict(H32=rel(f32["H"].cpu().numpy(), H_ref), S32=rel(la32.get_dopt_dhyper(), S_ref),
                   g32=rel(o32.grad(theta, w).cpu().numpy(), g_ref))
        c32 = np.linalg.cond(H_ref)
        for k, e in e32.items():
            worst[k] = max(worst.get(k, 0.0), e)
            if e > max(1e-4, 1e-7 * c32) or e != e:
                print("FAIL(fp32) case %d: N=%d D=%d %s cond=%.1e -> %s" % (it, N, D, fam, c32, e32), flush=True)
    cond = np.linalg.cond(H_ref)
    tolS = max(1e-10, 1e-15 * cond * 50)
    bad = [k for k, e in errs.items() if e > (tolS if k == "S" else 1e-10) or e != e]
    for k, e in errs.items():
        worst[k] = max(worst.get(k, 0.0), e)
    if bad:
        print("FAIL case %d: N=%d D=%d %s strided=%s offset=%s cond=%.1e -> %s" % (it, N, D, fam, strided, offset is not None, cond, errs), flush=True)
    if it % 25 == 0:
        print("case %d ok so far, worst %s" % (it, {k: "%.1e" % e for k, e in worst.items()}), flush=True)
print("done: %d cases, worst relative errors %s" % (cases, {k: "%.1e" % e for k, e in worst.items()}))
