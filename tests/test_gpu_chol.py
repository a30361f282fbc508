"""The two variants of the potrf panel substitution (hand-scheduled LDS reads / plain C++) must agree bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(extra_env):
    env = dict(os.environ)
    env.pop("VT_POTRF_C_PANEL", None)
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(HERE, "_potrf_worker.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l and l[0].isdigit()]


def test_potrf_panel_asm_variant_matches_the_c_loop_bitwise():
    """the asm variant of k_potrf_pack_small relies on LDS reads staying in flight across asm
    statements (s_waitcnt lgkmcnt(14 - m)); its results are compared with the same arithmetic written as a plain
    loop (VT_POTRF_C_PANEL=1), digest of the factor and of a solve, D = 16..128, so a toolchain change that breaks
    the assumption fails here."""
    a = _run({})
    b = _run({"VT_POTRF_C_PANEL": "1"})
    assert len(a) == 8 and a == b
