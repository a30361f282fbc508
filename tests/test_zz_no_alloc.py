"""vt_glm_hvp_noalloc (the one entry point that allocates nothing: its partial slabs live in a static
device array) against vt_glm_hvp (workspace argument), through the C-ABI, bit for bit, on two calls in
a row -- the second with a smaller input, so stale slabs of the first call would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(lib, torch, X, d, v, prior_scalar, prior=None):
    from vittles_amd import _lib
    N, D = X.shape
    ldx = X.stride(0) if N > 1 else D
    st = torch.cuda.current_stream().cuda_stream
    out_ws = torch.full((D,), float("nan"), dtype=torch.float64, device="cuda")
    out_na = torch.full((D,), float("nan"), dtype=torch.float64, device="cuda")
    nb = lib.vt_rows_workspace_bytes(D)
    ws = torch.empty(nb // 8 + 1, dtype=torch.float64, device="cuda")
    pp = 0 if prior is None else prior.data_ptr()
    _lib.check(lib.vt_glm_hvp(X.data_ptr(), ldx, d.data_ptr(), pp, prior_scalar, v.data_ptr(), N, D,
                              out_ws.data_ptr(), ws.data_ptr(), nb, st), "vt_glm_hvp")
    _lib.check(lib.vt_glm_hvp_noalloc(X.data_ptr(), ldx, d.data_ptr(), pp, prior_scalar, v.data_ptr(), N, D,
                                      out_na.data_ptr(), st), "vt_glm_hvp_noalloc")
    torch.cuda.synchronize()
    return out_ws.cpu().numpy(), out_na.cpu().numpy()


def test_noalloc_hvp_matches_workspace_hvp_two_calls_in_a_row():
    import torch
    from vittles_amd import _lib, device as dev
    lib = dev.require_gpu()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    # first a large input (wide-row kernel, every CU writes a slab), then smaller ones (fewer, shorter slabs)
    for (N, D, prior_scalar, use_prior_vec) in [(3000, 20000, 1.0, False), (1500, 333, 0.5, True),
                                                (40, 7, 0.0, False), (100000, 100, 2.0, False), (5, 4096, 1.0, False)]:
        X = torch.randn((N, D), dtype=torch.float64, device="cuda", generator=gen)
        d = torch.rand(N, dtype=torch.float64, device="cuda", generator=gen)
        v = torch.randn(D, dtype=torch.float64, device="cuda", generator=gen)
        prior = torch.rand(D, dtype=torch.float64, device="cuda", generator=gen) if use_prior_vec else None
        a, b = _both(lib, torch, X, d, v, prior_scalar, prior)
        assert np.all(np.isfinite(b))
        assert np.array_equal(a, b), (N, D, float(np.abs(a - b).max()))
        # and both against a NumPy HVP (tolerance of the parity bar: 1e-10 relative)
        Xh, dh, vh = X.cpu().numpy(), d.cpu().numpy(), v.cpu().numpy()
        ref = Xh.T @ (dh * (Xh @ vh)) + (prior.cpu().numpy() if prior is not None else prior_scalar) * vh
        assert np.abs(b - ref).max() / np.abs(ref).max() < 1e-10


def test_noalloc_rejects_sizes_beyond_its_scratch():
    import torch
    from vittles_amd import _lib, device as dev
    lib = dev.require_gpu()
    D = 20449
    X = torch.zeros((2, D), dtype=torch.float64, device="cuda")
    d = torch.ones(2, dtype=torch.float64, device="cuda")
    v = torch.ones(D, dtype=torch.float64, device="cuda")
    out = torch.empty(D, dtype=torch.float64, device="cuda")
    rc = lib.vt_glm_hvp_noalloc(X.data_ptr(), D, d.data_ptr(), 0, 0.0, v.data_ptr(), 2, D, out.data_ptr(), 0)
    assert rc == -6
    assert b"scratch" in lib.vt_error_string(rc)
    with pytest.raises(_lib.VittlesHipError, match="scratch"):
        _lib.check(rc, "vt_glm_hvp_noalloc")
