"""The RCCL code path on the one GPU a test box has: a world of ONE rank with the `nccl` backend
(= RCCL on ROCm) runs the same calls the row-sharded path makes on 2/4/8 GPUs -- the in-place sum
all-reduce of the [H | g] buffer, the all-reduced objective value and HVP, the broadcast that keeps
the optimiser's control flow rank-consistent -- and must reproduce the unsharded results bit for bit.
(World size 2 is covered on CPU with gloo in tests/test_distributed_cpu.py and on one card by
`VITTLES_BENCH_BACKEND=gloo torchrun ... bench.py`.)"""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_world_of_one_matches_unsharded(relerr):
    import torch
    import torch.distributed as dist
    import vittles_amd as va
    from vittles_amd.cg import GLMHessianVectorProduct
    from oracle import glm
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        N, D = 20000, 100
        X, y, w, _ = glm.synth(N, D, "logistic", seed=21)
        plain = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0)
        shard = va.WeightedGLMObjective(X, y, family="logistic", prior=1.0, process_group=dist.group.WORLD)
        theta_ref = glm.newton_optimum("logistic", X, y, w, prior=1.0)
        theta, info = va.optim.newton_glm(shard, w)                  # all-reduces + broadcasts every iteration
        assert info["converged"] and relerr(theta, theta_ref) < 1e-10
        a, b = plain.fused(theta, w), shard.fused(theta, w)
        assert torch.equal(a["H"], b["H"]) and torch.equal(a["g"], b["g"])
        assert plain(theta, w) == shard(theta, w)
        v = np.random.default_rng(0).standard_normal(D)
        assert np.array_equal(GLMHessianVectorProduct(plain, theta, w)(v), GLMHessianVectorProduct(shard, theta, w)(v))
        S = va.HyperparameterSensitivityLinearApproximation(shard, theta, w).get_dopt_dhyper()
        _, _, S_ref = glm.sensitivity("logistic", X, y, theta, w, prior=1.0)
        assert relerr(S, S_ref) < 1e-10
    finally:
        dist.destroy_process_group()


def test_c_abi_rccl_binding_world_of_one():
    """vt_comm_unique_id / vt_comm_init / vt_allreduce_sum / vt_comm_destroy from a torch-free host process (HIP
    runtime through ctypes, like a C caller): a world of one rank on GPU 0 (tests/_comm_worker.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-u", os.path.join(root, "tests", "_comm_worker.py")], capture_output=True,
                         text=True, timeout=240)
    assert res.returncode == 0, (res.stdout + res.stderr)[-3000:]
    if "comm unsupported" in res.stdout:
        pytest.skip("librccl.so is not on this box")
    assert "comm ok" in res.stdout
