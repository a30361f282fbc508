"""Child process of tests/test_gpu_chol.py: factor + pack a few SPD matrices and print a digest of the raw bytes of the
factor and of a solve through the TRSM operand pack (VT_POTRF_C_PANEL in the environment picks the panel-substitution variant)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from vittles_amd.linalg import DeviceCholesky

for D in (16, 17, 33, 64, 100, 101, 127, 128):
    rng = np.random.default_rng(D)
    A = rng.standard_normal((3 * D, D))
    H = torch.tensor(A.T @ A + np.eye(D), device="cuda")
    chol = DeviceCholesky.factor(H, pack=True)
    torch.cuda.synchronize()
    h = hashlib.sha256(chol.L.cpu().numpy().tobytes())
    # the pack buffer has unwritten padding: digest what it produces instead (the solve of a fixed right-hand side)
    h.update(chol.solve_rows(torch.tensor(A[:64].copy(), device="cuda")).cpu().numpy().tobytes())
    print(D, h.hexdigest(), flush=True)
