"""Drop-in for ``vittles/sparse_hessian_lib.py`` (reference lines 11-168): block-sparse
Hessians from M coloured Hessian-vector products, assembled on the MI355X.

Same constructor and methods; the result is a ``scipy.sparse.coo_matrix`` whose stored
entries are in exactly the reference's order (``ib`` outer, block middle, row-in-block inner;
global rows with both symmetric entries and the two 0.5*h halves for global pairs).  The HVPs
and the O(G*M) gather run as HIP kernels; the matrix additionally carries its device-side
block / border / corner images so that ``solver_lib.get_cholesky_solver`` can solve with the
batched block-arrowhead kernels instead of a generic sparse LU.

``objective_function`` must expose ``hvp(opt_par, v)`` on CUDA tensors (a registered
objective): the reference builds that HVP with autograd (``_append_jvp(grad(f))``,
sparse_hessian_lib.py:59-60), which is not part of this build.
"""
import numpy as np
import scipy.sparse
import torch
from scipy.sparse import coo_matrix

from . import _lib, device as dev


class SparseBlockHessian():
    def __init__(self, objective_function, sparsity_array):
        # reference :30-60
        self._fun = objective_function
        self._sparsity_array = np.asarray(sparsity_array)
        self._num_blocks = self._sparsity_array.shape[0]
        self._block_size = self._sparsity_array.shape[1]

        if len(np.unique(sparsity_array)) != len(self._sparsity_array.flatten()):
            raise ValueError(
                'The indices in ``sparsity array`` must be unique.')

        if not hasattr(objective_function, 'hvp'):
            raise NotImplementedError(
                'objective_function must be a registered objective exposing hvp(opt_par, v) '
                '(autograd is not part of this build).')
        self._inds_dev = None

    # -- helpers -------------------------------------------------------------------------
    def _device(self):
        return getattr(self._fun, 'device', torch.device('cuda', torch.cuda.current_device()))

    def _inds(self):
        if self._inds_dev is None:
            self._inds_dev = torch.as_tensor(
                np.ascontiguousarray(self._sparsity_array, dtype=np.int64)).to(self._device())
        return self._inds_dev

    def _hess_summed_term(self, opt_par, ib):
        """``ib`` is the index within the block (reference :62-67)."""
        lib = dev.require_gpu()
        v = torch.empty(opt_par.shape[0], dtype=torch.float64, device=opt_par.device)
        _lib.check(lib.vt_block_colour_vec(self._inds().data_ptr(), self._num_blocks, self._block_size, ib,
                                           v.data_ptr(), v.shape[0], dev.stream_ptr()), "vt_block_colour_vec")
        return self._fun.hvp(opt_par, v)

    def _block_parts(self, opt_par, print_every=0):
        lib = dev.require_gpu()
        G, M = self._num_blocks, self._block_size
        devc = opt_par.device
        nnz = G * M * M
        vals = torch.empty(nnz, dtype=torch.float64, device=devc)
        rows = torch.empty(nnz, dtype=torch.int64, device=devc)
        cols = torch.empty(nnz, dtype=torch.int64, device=devc)
        blocks = torch.empty((G, M, M), dtype=torch.float64, device=devc)
        for ib in range(M):
            if print_every > 0:
                if ib % print_every == 0:
                    print('Block index {} of {}.'.format(ib, M))
            hess_prod = self._hess_summed_term(opt_par, ib)
            _lib.check(lib.vt_block_gather(hess_prod.data_ptr(), self._inds().data_ptr(), G, M, ib,
                                           vals.data_ptr(), rows.data_ptr(), cols.data_ptr(),
                                           blocks.data_ptr(), dev.stream_ptr()), "vt_block_gather")
        if print_every > 0:
            print('Done differentiating.')
        return vals, rows, cols, blocks

    def get_block_hessian(self, opt_par, print_every=0):
        """Block Hessian at ``opt_par`` as a ``scipy.sparse.coo_matrix`` (reference :69-108)."""
        opt_np = np.atleast_1d(opt_par) if not isinstance(opt_par, torch.Tensor) else opt_par
        if opt_np.ndim != 1:
            raise ValueError('``opt_par`` must be a vector.')
        opt = dev.to_device(opt_par, self._device()).contiguous()
        vals, rows, cols, blocks = self._block_parts(opt, print_every)
        d = opt.shape[0]
        h_sparse = coo_matrix((vals.cpu().numpy(), (rows.cpu().numpy(), cols.cpu().numpy())), (d, d))
        _attach(h_sparse, ArrowheadStructure(self._inds(), blocks, None, None, None, d))
        return h_sparse

    def _global_parts(self, opt, global_inds, print_every=0):
        lib = dev.require_gpu()
        devc = opt.device
        G, M = self._num_blocks, self._block_size
        nloc, Kg = G * M, len(global_inds)
        local = self._inds().reshape(-1)
        ginds = torch.as_tensor(np.ascontiguousarray(global_inds, dtype=np.int64)).to(devc)
        per = 2 * nloc + 2 * Kg
        vals = torch.empty(Kg * per, dtype=torch.float64, device=devc)
        rows = torch.empty(Kg * per, dtype=torch.int64, device=devc)
        cols = torch.empty(Kg * per, dtype=torch.int64, device=devc)
        border = torch.empty((nloc, max(Kg, 1)), dtype=torch.float64, device=devc)
        corner = torch.empty((max(Kg, 1), max(Kg, 1)), dtype=torch.float64, device=devc)
        v = torch.zeros(opt.shape[0], dtype=torch.float64, device=devc)
        for count, ig in enumerate(global_inds):
            if print_every > 0:
                if count % print_every == 0:
                    print('Global index {} of {}.'.format(count, Kg))
            v[int(ig)] = 1
            hess_row = self._fun.hvp(opt, v)
            _lib.check(lib.vt_global_gather(hess_row.data_ptr(), local.data_ptr(), nloc, ginds.data_ptr(), Kg,
                                            count, vals.data_ptr(), rows.data_ptr(), cols.data_ptr(),
                                            border.data_ptr(), corner.data_ptr(), dev.stream_ptr()),
                       "vt_global_gather")
            v[int(ig)] = 0
        if print_every > 0:
            print('Done differentiating.')
        return vals, rows, cols, border, corner, ginds

    def get_global_hessian(self, opt_par, global_inds=None, print_every=0):
        """Dense Hessian terms of the global parameters (reference :110-163)."""
        local_inds = np.hstack(self._sparsity_array)
        if global_inds is None:
            global_inds = np.setdiff1d(np.arange(len(opt_par)), local_inds)

        global_local_intersection = np.intersect1d(global_inds, local_inds)
        if len(global_local_intersection) > 0:
            raise ValueError(
                'The global and local indices must be disjoint.  {}'.format(
                    global_local_intersection))
        opt = dev.to_device(opt_par, self._device()).contiguous()
        d = opt.shape[0]
        if len(global_inds) == 0:
            return coo_matrix((d, d))
        vals, rows, cols, border, corner, ginds = self._global_parts(opt, global_inds, print_every)
        h_sparse = coo_matrix((vals.cpu().numpy(), (rows.cpu().numpy(), cols.cpu().numpy())), (d, d))
        h_sparse._vt_global = (border, corner, ginds)
        return h_sparse

    def get_hessian(self, opt_par, print_every=0):
        # reference :165-168
        local_hessian = self.get_block_hessian(opt_par, print_every=print_every)
        global_hessian = self.get_global_hessian(opt_par, print_every=print_every)
        total = local_hessian + global_hessian
        struct = local_hessian._vt_structure
        glob = getattr(global_hessian, '_vt_global', None)
        if glob is not None:
            struct = ArrowheadStructure(struct.inds, struct.blocks, glob[0], glob[1], glob[2], struct.d)
        _attach(total, struct)
        return total


def _attach(mat, struct):
    mat._vt_structure = struct
    if struct.covers_everything():
        mat._vt_solver = struct.make_solver


class ArrowheadStructure:
    """Device images of H = [[blockdiag(B_g), C], [C^T, Gm]] and its batched solver (K8)."""

    def __init__(self, inds, blocks, border, corner, ginds, d):
        self.inds, self.blocks, self.border, self.corner, self.ginds, self.d = inds, blocks, border, corner, ginds, d
        self.G, self.M = blocks.shape[0], blocks.shape[1]
        self.Kg = 0 if ginds is None else int(ginds.shape[0])
        self._factored = None

    def covers_everything(self):
        return self.G * self.M + self.Kg == self.d

    def _factor(self):
        if self._factored is not None:
            return self._factored
        lib = dev.require_gpu()
        from .linalg import DeviceCholesky
        devc = self.blocks.device
        G, M, Kg = self.G, self.M, self.Kg
        L = self.blocks.clone()
        info = torch.zeros(1, dtype=torch.int32, device=devc)
        W = torch.empty((G * M, max(Kg, 1)), dtype=torch.float64, device=devc)
        spart = torch.empty((G, max(Kg * Kg, 1)), dtype=torch.float64, device=devc)
        schur = torch.empty((max(Kg, 1), max(Kg, 1)), dtype=torch.float64, device=devc)
        _lib.check(lib.vt_arrow_factor(L.data_ptr(), dev.ptr(self.border), dev.ptr(self.corner), W.data_ptr(),
                                       spart.data_ptr(), schur.data_ptr(), G, M, Kg, info.data_ptr(),
                                       dev.stream_ptr()), "vt_arrow_factor")
        k = int(info.item())
        if k > 0:
            raise np.linalg.LinAlgError("block Hessian is not positive definite (pivot %d)" % k)
        schur_chol = DeviceCholesky.factor(schur) if Kg > 0 else None
        self._factored = (L, W, schur_chol)
        return self._factored

    def make_solver(self):
        if not self.covers_everything():
            raise _lib.VittlesHipError("block structure does not cover the whole matrix")
        L, W, schur_chol = self._factor()
        lib = dev.require_gpu()
        G, M, Kg, d = self.G, self.M, self.Kg, self.d
        devc = L.device

        def solve(v):
            vd = dev.to_device(v, devc)
            one_d = vd.ndim == 1
            V = (vd.reshape(d, 1) if one_d else vd).contiguous()
            if V.shape[0] != d:
                raise ValueError("right-hand side must have %d rows" % d)
            nrhs = V.shape[1]
            x = torch.empty_like(V)
            tpart = torch.empty((G, max(Kg, 1), nrhs), dtype=torch.float64, device=devc)
            rhs_g = torch.empty((max(Kg, 1), nrhs), dtype=torch.float64, device=devc)
            _lib.check(lib.vt_arrow_local_solve(L.data_ptr(), dev.ptr(self.border), self.inds.data_ptr(),
                                                dev.ptr(self.ginds), V.data_ptr(), x.data_ptr(), tpart.data_ptr(),
                                                rhs_g.data_ptr(), G, M, Kg, nrhs, dev.stream_ptr()),
                       "vt_arrow_local_solve")
            if Kg > 0:
                rhs_g += V[self.ginds]
                xg = schur_chol.solve(rhs_g).contiguous()
                _lib.check(lib.vt_arrow_back(W.data_ptr(), self.inds.data_ptr(), self.ginds.data_ptr(),
                                             xg.data_ptr(), x.data_ptr(), G, M, Kg, nrhs, dev.stream_ptr()),
                           "vt_arrow_back")
            out = x.reshape(d) if one_d else x
            return dev.like_input(out, v)
        return solve
