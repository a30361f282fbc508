"""GPU counterpart of the sharding test: R row shards on ONE device through the HIP kernels --
partial Hessians (prior on shard 0 only) sum to the unsharded Hessian, and each shard's
sensitivities are the matching rows of the unsharded result (the RCCL all-reduce itself is
exercised by bench.py --gpus N on a multi-GPU node)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_shards_sum_to_unsharded(relerr):
    import torch
    import vittles_amd as va
    from vittles_amd import sharding
    from vittles_amd.linalg import DeviceCholesky
    from oracle import glm
    N, D, R = 5003, 100, 4
    X, y, w, _ = glm.synth(N, D, "logistic", seed=21)
    theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)
    Xd = torch.tensor(X, device="cuda")
    H = torch.zeros((D, D), dtype=torch.float64, device="cuda")
    parts = []
    for r in range(R):
        a, b = sharding.shard_rows(N, r, R)
        obj = va.WeightedGLMObjective(Xd[a:b], y[a:b], family="logistic", prior=1.0 if r == 0 else None)
        out = obj.fused(theta, w[a:b])
        H += out["H"]
        parts.append((a, b, out["r"]))
    H_ref, _, S_ref = glm.sensitivity("logistic", X, y, theta, w, prior=1.0)
    assert relerr(H.cpu().numpy(), H_ref) < 1e-10
    chol = DeviceCholesky.factor(H, pack=True)
    for a, b, r in parts:
        S = chol.solve_rows(Xd[a:b], rowscale=r, alpha=-1.0)
        assert relerr(S.cpu().numpy().T, S_ref[:, a:b]) < 1e-10 * np.abs(S_ref).max() / np.abs(S_ref[:, a:b]).max() + 1e-10
