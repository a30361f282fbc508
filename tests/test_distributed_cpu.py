"""CPU tests (gloo, world_size 2) of the row-sharded path: shard ranges, the single
all-reduce of (H, g), prior contributed once, sensitivities staying sharded.  The per-shard
arithmetic is the oracle's here (no GPU in this container); the GPU-side counterpart is
tests/test_gpu_sharded.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, N, D, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vittles_amd import sharding
        from oracle import glm
        import scipy.linalg
        X, y, w, _ = glm.synth(N, D, "logistic", seed=11)
        theta = glm.newton_optimum("logistic", X, y, w, prior=1.0)
        a, b = sharding.shard_rows(N, rank, world)
        Xs, ys, ws = X[a:b], y[a:b], w[a:b]
        prior = 1.0 if sharding.contributes_prior(dist.group.WORLD) else None
        H = torch.tensor(glm.hessian("logistic", Xs, ys, theta, ws, prior=prior))
        g = torch.tensor(glm.grad("logistic", Xs, ys, theta, ws, prior=prior))
        sharding.all_reduce_sum([H, g], dist.group.WORLD)
        jac_local = glm.cross_hessian("logistic", Xs, ys, theta, ws)
        S_local = -scipy.linalg.cho_solve(scipy.linalg.cho_factor(H.numpy()), jac_local)
        H_full, _, S_full = glm.sensitivity("logistic", X, y, theta, w, prior=1.0)
        # host-side control flow takes rank 0's numbers on every rank
        assert sharding.agree_scalars([float(rank) + 0.5, 2.0], dist.group.WORLD, None) == [0.5, 2.0]
        q.put((rank, float(np.abs(H.numpy() - H_full).max() / np.abs(H_full).max()),
               float(np.abs(S_local - S_full[:, a:b]).max() / np.abs(S_full).max()),
               float(np.linalg.norm(g.numpy())), (a, b)))
    finally:
        dist.destroy_process_group()


def test_shard_rows_partition():
    from vittles_amd import sharding
    for N in (0, 1, 7, 1000, 10**6 + 3):
        for R in (1, 2, 3, 8):
            rng = [sharding.shard_rows(N, r, R) for r in range(R)]
            assert rng[0][0] == 0 and rng[-1][1] == N
            assert all(rng[i][1] == rng[i + 1][0] for i in range(R - 1))
            assert max(b - a for a, b in rng) - min(b - a for a, b in rng) <= 1
    with pytest.raises(ValueError):
        sharding.shard_rows(10, 2, 2)


def test_row_sharded_world2_gloo():
    world, N, D = 2, 2001, 9
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, N, D, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert [r[4] for r in res] == [(0, 1000), (1000, 2001)]
    for rank, eh, es, gn, _ in res:
        assert eh < 1e-13 and es < 1e-12, (rank, eh, es)
        assert gn < 1e-8          # all-reduced gradient vanishes at the optimum on every rank
