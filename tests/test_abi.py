"""CPU tests (no GPU): the C-ABI library loads and exports every symbol the header declares,
and the host-side logic that runs before any kernel (argument checks, error strings)."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "vittles_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vittles_amd import _lib
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "libvittles_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "%s has no ctypes prototype in vittles_amd/_lib.py" % s
    assert set(_lib.SIGNATURES) == set(syms)


def test_version_and_error_strings():
    from vittles_amd import _lib
    lib = _lib.load()
    assert lib.vt_version() >= 100
    assert lib.vt_error_string(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert len(lib.vt_error_string(code)) > 3
    with pytest.raises(_lib.VittlesHipError):
        _lib.check(-2, "probe")


def test_workspace_queries_need_no_gpu():
    from vittles_amd import _lib
    lib = _lib.load()
    assert lib.vt_trsm_pack_bytes(100) == 2 * 28 * 256 * 8
    # mid range: at least the NB (NB + 1) chain-order images and the NB inverted diagonal blocks
    assert lib.vt_trsm_pack_bytes(129) >= (9 * 10 + 9) * 256 * 8
    assert lib.vt_trsm_pack_bytes(256) >= (16 * 17 + 16) * 256 * 8
    # the one-pass chain reaches D = 384 (NB = 24)
    assert lib.vt_trsm_pack_bytes(257) >= (17 * 18 + 17) * 256 * 8
    assert lib.vt_trsm_pack_bytes(384) >= (24 * 25 + 24) * 256 * 8
    # beyond: L and L^T images with the even leading dimension 386, and the chain packs of two 384-column blocks
    assert lib.vt_trsm_pack_bytes(385) >= (2 * 385 * 386 + 2 * (24 * 25 + 24) * 256) * 8
    assert lib.vt_trsm_pack_bytes(8193) == 0         # outside what the dense kernels cover
    assert lib.vt_potrf_workspace_bytes(100) >= 0


def test_public_names_match_reference_namespace():
    """vittles/__init__.py:1-8 of the reference."""
    import vittles_amd as va
    for name in ("HyperparameterSensitivityLinearApproximation", "solver_lib"):
        assert hasattr(va, name)
    for fn in ("get_dense_cholesky_solver", "get_sparse_cholesky_solver", "get_cholesky_solver", "get_cg_solver"):
        assert callable(getattr(va.solver_lib, fn))


def test_host_checks_before_any_kernel():
    import vittles_amd as va
    with pytest.raises(ValueError, match="must be sparse"):      # solver_lib.py:46-47
        va.solver_lib.get_sparse_cholesky_solver(np.eye(3))
    with pytest.raises(NotImplementedError):                      # no autograd in this build
        va.HyperparameterSensitivityLinearApproximation(lambda t, w: 0.0, np.zeros(2), np.zeros(3))


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import vittles_amd as va
    from vittles_amd import _lib
    with pytest.raises(_lib.VittlesHipError, match="no CPU fallback"):
        va.WeightedGLMObjective(np.zeros((4, 2)), np.zeros(4))
    with pytest.raises(_lib.VittlesHipError):
        va.solver_lib.get_dense_cholesky_solver(np.eye(3))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "vittles_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_next_rows_host_checks_need_no_gpu():
    """Host logic of the SURVEY.md section 8(f) rows that runs before any kernel."""
    import vittles_amd as va
    from vittles_amd import _lib
    lib = _lib.load()
    assert lib.vt_gemm_tn_workspace_bytes(64, 100) > 0
    assert lib.vt_gemm_tn_workspace_bytes(64, 100) % (128 * 128 * 8) == 0     # whole 128 x 128 slabs
    assert lib.vt_gemm_tn_workspace_bytes(8193, 10) == 0 and lib.vt_gemm_tn_workspace_bytes(0, 10) == 0
    with pytest.raises(TypeError):
        va.optim.newton_glm(lambda t, w: 0.0, np.ones(3))
    with pytest.raises(NotImplementedError):                      # arbitrary callables need autograd
        va.ParametricSensitivityTaylorExpansion.optimization_objective(
            lambda t, w: 0.0, np.zeros(2), np.zeros(3), order=2)
    assert isinstance(va.__version__, str)
