"""The row-sharded PRODUCT path with two real ranks: two fresh processes
share GPU 0 over gloo and run WeightedGLMObjective(process_group=) -> H, g, newton_glm, the
drop-in sensitivity class (sharded S, all-reduced predictions, VJP / JVP), HVP + CG and the
Taylor terms against the oracle on the unsharded data (tests/_rank_worker.py); and
`bench.py --gpus 2` starts its own two ranks and reports n_gpus = 2."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_product_path_vs_oracle():
    world, port = 2, _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "_rank_worker.py")],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-4000:])
        assert "rank %d ok" % r in o


def test_bench_gpus2_spawns_two_ranks():
    env = dict(os.environ, VITTLES_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5",
                          "--warmup", "1", "--rows", "50000", "--c3-rows", "40000", "--c3-dim", "300",
                          "--c4-rows", "4000", "--hvp-dim", "5000"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks"] == 2
    assert out["rccl_ranks_seen"] is None and out["config"]["collective_backend"] == "gloo"   # no RCCL in a gloo rehearsal
    assert len(out["per_rank_ms_per_step"]) == 2
    assert out["stages_ms"]["all_reduce_H_g"] > 0.0
    assert out["value"] > 0 and out["scaling"] == "weak"
    # the rank-symmetric legs of configs[2] (strong-scaled dense path) and configs[3] (CG with an all-reduce per iteration)
    c3 = out["config3_strong"]
    assert c3["n_gpus"] == 2 and c3["rows_per_gpu"] == 20000 and c3["rows_total"] == 40000
    assert c3["stages_ms"]["all_reduce_H_g"] > 0.0 and c3["parity"]["H_identical_on_all_ranks"]
    for key in ("S_sampled_rows_rel_err_max_over_ranks", "checksum_all_rows_rel_err", "H_first_rows_rel_err_vs_numpy"):
        assert c3["parity"][key] < 1e-9, key
    assert c3["parity"]["regenerated_rows_max_abs_diff"] < 1e-12
    c4 = out["config4_strong"]
    assert c4["n_gpus"] == 2 and c4["cg_info"] == 0 and c4["true_residual_rel"] < 1e-4
    assert c4["all_reduce_D_vector_ms"] > 0.0 and c4["cg_iterations"] > 0
