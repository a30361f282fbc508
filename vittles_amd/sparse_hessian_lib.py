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

from . import _lib, device as dev, sharding


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
        self._host_cache = None

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

    def _block_parts(self, opt_par, print_every=0, out=None):
        """COO triplets (reference order) + dense block images of the block part, on the device.  ``out``:
        optional (vals, rows, cols) views of a larger buffer to fill (get_hessian packs both parts in one)."""
        lib = dev.require_gpu()
        G, M = self._num_blocks, self._block_size
        devc = opt_par.device
        nnz = G * M * M
        if out is None:
            out = (torch.empty(nnz, dtype=torch.float64, device=devc),
                   torch.empty(nnz, dtype=torch.int64, device=devc),
                   torch.empty(nnz, dtype=torch.int64, device=devc))
        vals, rows, cols = out
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
        h_sparse = coo_matrix(_triplets_to_host(vals, rows, cols, d), (d, d))
        _attach(h_sparse, ArrowheadStructure(self._inds(), blocks, None, None, None, d))
        return h_sparse

    def _global_parts(self, opt, global_inds, print_every=0, out=None):
        lib = dev.require_gpu()
        devc = opt.device
        G, M = self._num_blocks, self._block_size
        nloc, Kg = G * M, len(global_inds)
        local = self._inds().reshape(-1)
        ginds = torch.as_tensor(np.ascontiguousarray(global_inds, dtype=np.int64)).to(devc)
        per = 2 * nloc + 2 * Kg
        if out is None:
            out = (torch.empty(Kg * per, dtype=torch.float64, device=devc),
                   torch.empty(Kg * per, dtype=torch.int64, device=devc),
                   torch.empty(Kg * per, dtype=torch.int64, device=devc))
        vals, rows, cols = out
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
        h_sparse = coo_matrix(_triplets_to_host(vals, rows, cols, d), (d, d))
        h_sparse._vt_global = (border, corner, ginds)
        return h_sparse

    def get_hessian(self, opt_par, print_every=0):
        """Block + global Hessian (reference :165-168 returns ``local_hessian + global_hessian``, which SciPy
        sums into a CSR matrix on the host).  Here the triplets of BOTH parts are gathered into one device
        buffer in the reference's order (block part first) and cross the bus once; the result is a single
        ``coo_matrix`` holding exactly the entries the reference adds up (duplicates -- the two 0.5*h halves of
        a global pair -- are summed by SciPy on conversion, as in ``+``)."""
        opt_np = np.atleast_1d(opt_par) if not isinstance(opt_par, torch.Tensor) else opt_par
        if opt_np.ndim != 1:
            raise ValueError('``opt_par`` must be a vector.')
        opt = dev.to_device(opt_par, self._device()).contiguous()
        d = opt.shape[0]
        devc = opt.device
        # the default global indices (reference :114-116) and the (row, col) arrays of the result depend on the
        # sparsity array and d only: computed once, kept on the host
        cache = self._host_cache if getattr(self, '_host_cache', None) and self._host_cache['d'] == d else None
        if cache is None:
            global_inds = np.setdiff1d(np.arange(d), np.hstack(self._sparsity_array))
            cache = self._host_cache = dict(d=d, global_inds=global_inds, ij=None)
        global_inds = cache['global_inds']
        G, M = self._num_blocks, self._block_size
        Kg = len(global_inds)
        nb = G * M * M
        ng = Kg * (2 * G * M + 2 * Kg)
        vals = torch.empty(nb + ng, dtype=torch.float64, device=devc)
        rows = torch.empty(nb + ng, dtype=torch.int64, device=devc)
        cols = torch.empty(nb + ng, dtype=torch.int64, device=devc)
        _, _, _, blocks = self._block_parts(opt, print_every, out=(vals[:nb], rows[:nb], cols[:nb]))
        group = getattr(self._fun, 'process_group', None)
        struct = ArrowheadStructure(self._inds(), blocks, None, None, None, d, process_group=group)
        if Kg > 0:
            _, _, _, border, corner, ginds = self._global_parts(opt, global_inds, print_every,
                                                                out=(vals[nb:], rows[nb:], cols[nb:]))
            struct = ArrowheadStructure(self._inds(), blocks, border, corner, ginds, d, process_group=group)
        if cache['ij'] is None:
            hv, ij = _triplets_to_host(vals, rows, cols, d)
            cache['ij'] = (ij[0].copy(), ij[1].copy())
        else:                                   # only the values cross the bus; every result owns its index arrays
            hv, ij = vals.cpu().numpy(), (cache['ij'][0].copy(), cache['ij'][1].copy())
        total = coo_matrix((hv, ij), (d, d))
        _attach(total, struct)
        return total


def _triplets_to_host(vals, rows, cols, d):
    """(vals f64, rows, cols) device tensors -> numpy arrays in ONE device-to-host copy; indices travel as int32
    when they fit (SciPy's own index type for such matrices)."""
    n = vals.shape[0]
    if d < 2 ** 31:
        packed = torch.empty(2 * n, dtype=torch.float64, device=vals.device)
        packed[:n] = vals
        idx = packed[n:].view(torch.int32)          # 2 n int32 slots: rows then cols
        idx[:n] = rows
        idx[n:] = cols
        host = packed.cpu().numpy()
        ij = host[n:].view(np.int32)
        return host[:n], (ij[:n], ij[n:])
    return vals.cpu().numpy(), (rows.cpu().numpy(), cols.cpu().numpy())


def matrix_fingerprint(mat):
    """Check that a scipy sparse matrix still holds the content a device-side structure snapshot was taken from
    (an in-place edit of ``.data`` or of the index arrays keeps Python attributes alive): shape, nnz and a CRC of
    the values AND the index arrays -- O(nnz), and unlike sums it sees entries that were swapped or moved."""
    import zlib
    crc = zlib.crc32(np.ascontiguousarray(mat.data).tobytes())
    for name in ("row", "col", "indices", "indptr"):
        arr = getattr(mat, name, None)
        if arr is not None:
            crc = zlib.crc32(np.ascontiguousarray(arr).tobytes(), crc)
    return (tuple(mat.shape), int(mat.nnz), type(mat).__name__, crc)


SYM_RTOL = 1e-12


def _require_symmetric(a, b, what):
    """``a`` must equal ``b`` (its mirror image) to rounding: the device solvers are Cholesky solvers and read one
    triangle; the reference's SuperLU (vittles/solver_lib.py:48) would silently solve a different matrix."""
    scale = max(float(np.abs(a).max(initial=0.0)), float(np.abs(b).max(initial=0.0)))
    if scale > 0.0 and float(np.abs(a - b).max()) > SYM_RTOL * scale:
        raise ValueError(
            'get_sparse_cholesky_solver needs a SYMMETRIC (positive definite) matrix with both triangles stored; '
            'the %s is not symmetric (max |h - h^T| = %.3g)' % (what, float(np.abs(a - b).max())))


def _attach(mat, struct):
    mat._vt_structure = struct
    mat._vt_fingerprint = matrix_fingerprint(mat)
    if struct.covers_everything():
        mat._vt_solver = struct.make_solver


MAX_BLOCK = 64      # vt_arrow_* kernels: block size M and number of globals Kg


def detect_arrowhead(h, sparsity_array=None, global_inds=None):
    """Recover the block-arrowhead structure H = [[blockdiag(B_g), C], [C^T, Gm]] of a symmetric scipy
    sparse matrix from its sparsity pattern (or take it from explicit ``sparsity_array`` (G, M) / ``global_inds``).

    Global indices: rows whose degree exceeds what a local row can have (M + Kg <= 2 * MAX_BLOCK); if a
    connected component of what remains is still larger than MAX_BLOCK, its highest-degree node joins the
    globals, and so on (at most MAX_BLOCK globals).  Blocks: the connected components of the rest.  They need not
    be index ranges nor of one size: components are packed (first fit, largest first) into G bins of
    M = largest component, and a bin that is not full is topped up with PADDING indices d, d+1, ... (decoupled
    unknowns with a unit diagonal that the solver adds and drops again).

    Returns (inds (G, M) int64, global_inds (Kg,) int64, d_pad) or None when the matrix is not of that form."""
    import scipy.sparse.csgraph
    d = h.shape[0]
    if sparsity_array is not None:
        inds = np.asarray(sparsity_array, dtype=np.int64)
        if inds.ndim != 2 or len(np.unique(inds)) != inds.size:
            raise ValueError('The indices in ``sparsity array`` must be unique.')
        if global_inds is None:
            global_inds = np.setdiff1d(np.arange(d), inds.reshape(-1))
        ginds = np.asarray(global_inds, dtype=np.int64)
        if len(np.intersect1d(ginds, inds.reshape(-1))) > 0:
            raise ValueError('The global and local indices must be disjoint.')
        if inds.size + ginds.size != d:
            return None
        return inds, ginds, d
    coo = scipy.sparse.coo_matrix(h)
    off = coo.row != coo.col
    r, c = coo.row[off], coo.col[off]
    degree = np.bincount(r, minlength=d) + 1
    local = np.ones(d, dtype=bool)
    if global_inds is None:
        local[degree > 2 * MAX_BLOCK] = False
    else:
        local[np.asarray(global_inds, dtype=np.int64)] = False
    for _ in range(MAX_BLOCK + 1):
        if d - int(local.sum()) > MAX_BLOCK:
            return None
        keep = local[r] & local[c]
        graph = scipy.sparse.coo_matrix((np.ones(int(keep.sum()), dtype=np.int8), (r[keep], c[keep])), (d, d))
        ncomp, labels = scipy.sparse.csgraph.connected_components(graph, directed=False)
        sizes = np.bincount(labels[local], minlength=ncomp)      # global nodes are singletons of size 0 here
        big = np.nonzero(sizes > MAX_BLOCK)[0]
        if big.size == 0:
            break
        if global_inds is not None:
            return None                                           # the caller named the globals: no guessing
        # the best-connected node of every oversized component becomes a global
        deg_local = np.bincount(r[keep], minlength=d)
        for comp in big:
            members = np.nonzero((labels == comp) & local)[0]
            local[members[np.argmax(deg_local[members])]] = False
    else:
        return None
    ginds = np.nonzero(~local)[0].astype(np.int64)
    lidx = np.nonzero(local)[0]
    if lidx.size == 0:
        return None
    # components (members in ascending index order), largest first
    lab = labels[lidx]
    order = np.argsort(lab, kind="stable")
    members_sorted = lidx[order]
    comp_ids, starts, counts = np.unique(lab[order], return_index=True, return_counts=True)
    M = int(counts.max())
    if np.all(counts == M):
        return members_sorted.reshape(-1, M).astype(np.int64), ginds, d
    # first-fit-decreasing into bins of capacity M; by size class this is a counting problem, so it is done per
    # size class instead of per component: fill every open bin with the largest remaining components that fit
    by_size = np.argsort(-counts, kind="stable")
    bins = []           # lists of member arrays
    room = []           # free slots per bin
    open_by_room = {}   # free slots -> stack of bin numbers
    for ci in by_size:
        sz = int(counts[ci])
        mem = members_sorted[starts[ci]:starts[ci] + sz]
        slot = None
        for free in range(sz, M):                 # tightest open bin that still fits (best fit)
            stack = open_by_room.get(free)
            if stack:
                slot = stack.pop()
                break
        if slot is None:
            bins.append([mem])
            room.append(M - sz)
            slot = len(bins) - 1
        else:
            bins[slot].append(mem)
            room[slot] -= sz
        if room[slot] > 0:
            open_by_room.setdefault(room[slot], []).append(slot)
    G = len(bins)
    inds = np.empty((G, M), dtype=np.int64)
    pad = d
    for g, parts in enumerate(bins):
        row = np.concatenate(parts)
        if row.size < M:
            npad = M - row.size
            row = np.concatenate([row, np.arange(pad, pad + npad)])
            pad += npad
        inds[g] = row
    return inds, ginds, pad


def arrowhead_from_sparse(h, device=None, sparsity_array=None, global_inds=None):
    """``ArrowheadStructure`` (device images of the blocks, border and corner) of a scipy sparse matrix whose
    pattern is block-arrowhead, or None.  O(nnz) host work, no densification."""
    found = detect_arrowhead(h, sparsity_array, global_inds)
    if found is None:
        return None
    inds, ginds, d_pad = found
    G, M = inds.shape
    Kg = int(ginds.size)
    if M > MAX_BLOCK or Kg > MAX_BLOCK:
        return None
    d = h.shape[0]
    coo = scipy.sparse.coo_matrix(h)
    comp = np.full(d_pad, -1, dtype=np.int64)   # block of a local index
    pos = np.zeros(d_pad, dtype=np.int64)       # position inside its block / among the globals
    comp[inds.reshape(-1)] = np.repeat(np.arange(G), M)
    pos[inds.reshape(-1)] = np.tile(np.arange(M), G)
    pos[ginds] = np.arange(Kg)
    r, c, v = coo.row, coo.col, np.asarray(coo.data, dtype=np.float64)
    rl, cl = comp[r] >= 0, comp[c] >= 0
    ll = rl & cl
    if np.any(comp[r[ll]] != comp[c[ll]]):
        return None                              # an entry couples two different blocks
    blocks = np.zeros((G, M, M))
    np.add.at(blocks, (comp[r[ll]], pos[r[ll]], pos[c[ll]]), v[ll])      # duplicates are summed (COO rule)
    if d_pad > d:                                # padding unknowns: unit diagonal, coupled to nothing
        padi = np.arange(d, d_pad)
        blocks[comp[padi], pos[padi], pos[padi]] = 1.0
    _require_symmetric(blocks, blocks.transpose(0, 2, 1), 'block diagonal')
    border = corner = None
    if Kg > 0:
        border = np.zeros((G * M, Kg))
        lg = rl & ~cl
        np.add.at(border, (comp[r[lg]] * M + pos[r[lg]], pos[c[lg]]), v[lg])
        mirror = np.zeros((G * M, Kg))           # the (global row, local column) entries: must mirror the border
        gl = ~rl & cl
        np.add.at(mirror, (comp[c[gl]] * M + pos[c[gl]], pos[r[gl]]), v[gl])
        _require_symmetric(border, mirror, 'border (local x global against global x local entries)')
        corner = np.zeros((Kg, Kg))
        gg = ~rl & ~cl
        np.add.at(corner, (pos[r[gg]], pos[c[gg]]), v[gg])
        _require_symmetric(corner, corner.T, 'global corner')
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device())
    to = lambda x: None if x is None else torch.as_tensor(x).to(device)
    return ArrowheadStructure(to(inds), to(blocks), to(border), to(corner), to(ginds) if Kg > 0 else None, d,
                              d_pad=d_pad)


class ArrowheadStructure:
    """Device images of H = [[blockdiag(B_g), C], [C^T, Gm]] and its batched solver (K8)."""

    def __init__(self, inds, blocks, border, corner, ginds, d, d_pad=None, process_group=None):
        self.inds, self.blocks, self.border, self.corner, self.ginds, self.d = inds, blocks, border, corner, ginds, d
        self.d_pad = d if d_pad is None else int(d_pad)    # d + padding unknowns (detect_arrowhead)
        # blocks sharded over the ranks of this group (each rank: its blocks + the replicated globals, the FULL
        # corner): the Schur complement and the globals' right-hand side are all-reduced (Kg x Kg, Kg x nrhs),
        # the block factorisations and local solves need no communication
        self.process_group = process_group
        self.G, self.M = blocks.shape[0], blocks.shape[1]
        self.Kg = 0 if ginds is None else int(ginds.shape[0])
        self._factored = None

    def covers_everything(self):
        return self.G * self.M + self.Kg == self.d_pad

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
        corner = self.corner
        group = self.process_group
        if group is not None and Kg > 0 and sharding.group_rank(group) != 0:
            corner = torch.zeros_like(self.corner)          # the corner enters the Schur complement once
        _lib.check(lib.vt_arrow_factor(L.data_ptr(), dev.ptr(self.border), dev.ptr(corner), W.data_ptr(),
                                       spart.data_ptr(), schur.data_ptr(), G, M, Kg, info.data_ptr(),
                                       dev.stream_ptr()), "vt_arrow_factor")
        k = int(info.item())
        if group is not None:
            (k,) = [int(v) for v in sharding.all_reduce_max([float(k)], group, devc)]
        if k > 0:
            raise np.linalg.LinAlgError("block Hessian is not positive definite (pivot %d)" % k)
        if group is not None and Kg > 0:
            sharding.all_reduce_sum(schur, group)           # Gm - sum over ALL ranks' groups of C_g^T B_g^-1 C_g
        schur_chol = DeviceCholesky.factor(schur) if Kg > 0 else None
        self._factored = (L, W, schur_chol)
        return self._factored

    def make_solver(self):
        if not self.covers_everything():
            raise _lib.VittlesHipError("block structure does not cover the whole matrix")
        L, W, schur_chol = self._factor()
        lib = dev.require_gpu()
        G, M, Kg, d, d_pad = self.G, self.M, self.Kg, self.d, self.d_pad
        devc = L.device

        def solve(v):
            vd = dev.to_device(v, devc)
            one_d = vd.ndim == 1
            if vd.shape[0] != d:
                raise ValueError("right-hand side must have %d rows" % d)
            V = (vd.reshape(d, 1) if one_d else vd).contiguous()
            nrhs = V.shape[1]
            if d_pad > d:                     # padding unknowns: zero right-hand side, dropped from the result
                Vp = torch.zeros((d_pad, nrhs), dtype=torch.float64, device=devc)
                Vp[:d] = V
                V = Vp
            x = torch.empty_like(V)
            tpart = torch.empty((G, max(Kg, 1), nrhs), dtype=torch.float64, device=devc)
            rhs_g = torch.empty((max(Kg, 1), nrhs), dtype=torch.float64, device=devc)
            _lib.check(lib.vt_arrow_local_solve(L.data_ptr(), dev.ptr(self.border), self.inds.data_ptr(),
                                                dev.ptr(self.ginds), V.data_ptr(), x.data_ptr(), tpart.data_ptr(),
                                                rhs_g.data_ptr(), G, M, Kg, nrhs, dev.stream_ptr()),
                       "vt_arrow_local_solve")
            if Kg > 0:
                sharding.all_reduce_sum(rhs_g, self.process_group)     # -sum over all ranks' groups of C_g^T y_g
                rhs_g += V[self.ginds]
                xg = schur_chol.solve(rhs_g).contiguous()
                _lib.check(lib.vt_arrow_back(W.data_ptr(), self.inds.data_ptr(), self.ginds.data_ptr(),
                                             xg.data_ptr(), x.data_ptr(), G, M, Kg, nrhs, dev.stream_ptr()),
                           "vt_arrow_back")
            x = x[:d]
            out = x.reshape(d) if one_d else x
            return dev.like_input(out, v)
        return solve
