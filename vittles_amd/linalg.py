"""Host wrappers over the dense device solvers (K4/K5): Cholesky factor object."""
import numpy as np
import torch

from . import _lib, device as dev


class DeviceCholesky:
    """H = L L^T on the device.  Mirrors what ``scipy.linalg.cho_factor`` /
    ``cho_solve`` give the reference (vittles/solver_lib.py:26-29)."""

    def __init__(self, L, pack=None, info=None):
        self.L = L                  # (D, D) CUDA fp64, lower triangle valid
        self.D = L.shape[0]
        self._pack = pack           # MFMA TRSM operand images (lazy)
        self._pack32 = None         # the same images in float for the fp32 storage mode (lazy)
        self._slab_ws = None        # double scratch of the fp32 mode's slab-wise solve at D > 128 (lazy)
        self._info = info           # device int32: LAPACK-style potrf info (None = known good)

    def raise_if_failed(self):
        """Read potrf's info (one host sync) and raise like SciPy does
        (``scipy/linalg/_decomp_cholesky.py:37-39``)."""
        if self._info is not None:
            k = int(self._info.item())
            self._info = None
            if k > 0:
                raise np.linalg.LinAlgError(
                    "%d-th leading minor of the array is not positive definite" % k)

    def info_to_host_async(self, host_slot):
        """Queue a copy of potrf's info into ``host_slot`` (a 1-element pinned int32 CPU tensor)
        and return an event that completes when it has landed: lets a pipeline of factorisations
        be checked one step late without draining the stream."""
        host_slot.copy_(self._info, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    @staticmethod
    def raise_for_info(k):
        if k > 0:
            raise np.linalg.LinAlgError(
                "%d-th leading minor of the array is not positive definite" % k)

    # -- construction -----------------------------------------------------
    @classmethod
    def factor(cls, H, check=True, pack=False):
        """Factor a (D, D) CUDA tensor (not modified).  Raises LinAlgError like SciPy
        when H is not positive definite; with ``check=False`` the test is deferred to
        ``raise_if_failed()`` so that dependent kernels can be queued without a sync."""
        lib = dev.require_gpu()
        D = H.shape[0]
        if H.ndim != 2 or H.shape[1] != D:
            raise ValueError('Input array needs to be a square matrix.')
        info = torch.empty(1, dtype=torch.int32, device=H.device)
        pk = None
        if pack and lib.vt_trsm_pack_bytes(D) > 0:
            # one launch, out of place: blocked Cholesky + the MFMA TRSM operand images
            Hs = H.detach()
            if Hs.stride(1) != 1:
                Hs = Hs.contiguous()
            L = torch.empty((D, D), dtype=torch.float64, device=H.device)
            pk = torch.empty(lib.vt_trsm_pack_bytes(D) // 8, dtype=torch.float64, device=L.device)
            _lib.check(lib.vt_potrf_pack_from(Hs.data_ptr(), Hs.stride(0), L.data_ptr(), D, D, info.data_ptr(),
                                              pk.data_ptr(), dev.stream_ptr()), "vt_potrf_pack_from")
        else:
            L = H.detach().clone().contiguous()
            nws = lib.vt_potrf_workspace_bytes(D)
            ws = torch.empty(nws // 8 + 1, dtype=torch.float64, device=L.device) if nws else None
            _lib.check(lib.vt_potrf(L.data_ptr(), D, D, info.data_ptr(), dev.ptr(ws), nws, dev.stream_ptr()),
                       "vt_potrf")
        chol = cls(L, pack=pk, info=info)
        if check:
            chol.raise_if_failed()
        return chol

    @classmethod
    def from_scipy(cls, h_chol, device=None):
        """Adopt a factor made by ``scipy.linalg.cho_factor`` -> ``(c, lower)``
        (vittles/solver_lib.py:15-18: if given, ``h`` is ignored)."""
        c, lower = h_chol
        c = dev.to_device(c, device)
        L = c if lower else c.t()
        return cls(torch.tril(L).contiguous())

    # -- solves -------------------------------------------------------------
    def solve(self, v):
        """H^{-1} v for v of shape (D,) or (D, K) (CUDA tensor); returns a new tensor."""
        lib = dev.require_gpu()
        if v.shape[0] != self.D:
            raise ValueError("incompatible dimensions: factor is %dx%d, b has %d rows"
                             % (self.D, self.D, v.shape[0]))
        # Every D the MFMA TRSM covers goes through it, one 16-row tile for a single right-hand side: the
        # generic one-wave substitution (vt_potrs) costs 0.1 ms at D = 100 and 4 ms at D = 1000, the tile
        # kernels 0.015 / 0.3 ms (the operand pack is built once per factor and cached).
        tiled = lib.vt_trsm_pack_bytes(self.D) > 0
        if v.ndim == 1:
            if tiled:
                return self.solve_rows(v.detach().reshape(1, self.D)).reshape(self.D)
            x = v.detach().clone().contiguous()
            _lib.check(lib.vt_potrs(self.L.data_ptr(), self.D, self.D, x.data_ptr(), 1, self.D, 1,
                                    dev.stream_ptr()), "vt_potrs")
            return x
        K = v.shape[1]
        if tiled:
            # MFMA TRSM wants right-hand sides as rows: (K, D) row-major.
            rows = v.detach().t().contiguous()
            out = self.solve_rows(rows)
            return out.t()          # (D, K), F-contiguous like cho_solve's result
        x = v.detach().clone().contiguous()
        _lib.check(lib.vt_potrs(self.L.data_ptr(), self.D, self.D, x.data_ptr(), K, 1, K,
                                dev.stream_ptr()), "vt_potrs")
        return x

    def pack(self):
        if self._pack is None:
            lib = dev.require_gpu()
            nbytes = lib.vt_trsm_pack_bytes(self.D)
            if nbytes == 0:
                raise _lib.VittlesHipError("MFMA TRSM pack supports D <= 8192 (got %d)" % self.D)
            self._pack = torch.empty(nbytes // 8, dtype=torch.float64, device=self.L.device)
            _lib.check(lib.vt_trsm_pack(self.L.data_ptr(), self.D, self.D, self._pack.data_ptr(),
                                        dev.stream_ptr()), "vt_trsm_pack")
        return self._pack

    def pack32(self):
        if self._pack32 is None:
            lib = dev.require_gpu()
            nbytes = lib.vt_trsm_pack_f32_bytes(self.D)
            if nbytes == 0:
                raise _lib.VittlesHipError("the fp32 mode's TRSM supports D <= 128 (got %d)" % self.D)
            pk = self.pack()
            self._pack32 = torch.empty(nbytes // 4, dtype=torch.float32, device=self.L.device)
            _lib.check(lib.vt_trsm_pack_f32(pk.data_ptr(), self.D, self._pack32.data_ptr(), dev.stream_ptr()),
                       "vt_trsm_pack_f32")
        return self._pack32

    def solve_rows(self, B, rowscale=None, alpha=1.0, out=None):
        """out[n, :] = alpha * rowscale[n] * H^{-1} B[n, :]  for a row-major (N, D) B.
        A float32 ``B`` (fp32 storage mode) is solved by the f32 MFMA chain into a float32 result; the factor
        itself is always fp64."""
        lib = dev.require_gpu()
        N, D = B.shape
        if D != self.D:
            raise ValueError("B must have %d columns" % self.D)
        if B.stride(1) != 1:
            B = B.contiguous()
        if out is None:
            # rows on a 16-byte pitch: the LDS-DMA kernels that read the result later (large-D updates, W^T S) need it
            # (sensitivity-sized results only: a handful of right-hand sides keeps the plain dense layout)
            per16 = 16 // B.element_size()
            Dp = (D + per16 - 1) // per16 * per16 if (N >= 4096 and D > 80) else D   # narrow: HBM-bound, dense rows
            full = torch.empty((N, Dp), dtype=B.dtype, device=B.device)
            if Dp > D:
                full[:, D:] = 0
            out = full[:, :D]
        # a single row's stride is arbitrary (torch reports 1 for a transposed (D, 1) column)
        ldb = B.stride(0) if N > 1 else D
        ldo = out.stride(0) if N > 1 else D
        if B.dtype == torch.float32:
            if out.dtype != torch.float32:
                raise ValueError("out must be float32 like B")
            if D > _lib.SMALL_D_MAX:
                # fp32 storage at D > 128: slabs of rows widened into a double scratch, solved by the fp64 kernels
                # with the ordinary pack and narrowed into the float result (csrc/vt_f32.hip)
                nbytes = lib.vt_trsm_rows_f32_large_workspace_bytes(N, D)
                if nbytes == 0:
                    raise _lib.VittlesHipError("the fp32 mode's solve supports D <= 8192 (got %d)" % D)
                ws = self._slab_ws
                if ws is None or ws.numel() * 8 < nbytes or ws.device != B.device:
                    ws = self._slab_ws = torch.empty(nbytes // 8, dtype=torch.float64, device=B.device)
                _lib.check(lib.vt_trsm_rows_f32_large(self.pack().data_ptr(), D, B.data_ptr(), ldb, dev.ptr(rowscale),
                                                      float(alpha), N, out.data_ptr(), ldo, ws.data_ptr(), ws.numel() * 8,
                                                      dev.stream_ptr()), "vt_trsm_rows_f32_large")
                return out
            _lib.check(lib.vt_trsm_rows_f32(self.pack32().data_ptr(), D, B.data_ptr(), ldb, dev.ptr(rowscale),
                                            float(alpha), N, out.data_ptr(), ldo, dev.stream_ptr()),
                       "vt_trsm_rows_f32")
            return out
        _lib.check(lib.vt_trsm_rows(self.pack().data_ptr(), D, B.data_ptr(), ldb,
                                    dev.ptr(rowscale), float(alpha), N, out.data_ptr(), ldo,
                                    dev.stream_ptr()), "vt_trsm_rows")
        return out
