"""ORACLE (test infrastructure, never a product path): restatement of
``/root/reference/vittles/sparse_hessian_lib.py`` with ``_f_fwd_hess`` (the
forward-over-reverse HVP built by ``_append_jvp``, sensitivity_lib.py:440-492 /
sparse_hessian_lib.py:59-60) replaced by an explicit ``hvp(opt_par, v)`` callable.

The COO entry ORDER and the duplicate-entry rules are kept exactly:
  * block part: ``ib`` outer, block ``b`` middle, row-in-block inner
    (sparse_hessian_lib.py:93-102);
  * global part: per global index, (il,ig)&(ig,il) for every local il
    (:137-144), then two 0.5*h halves for every global pair (:146-153).
"""
import numpy as np
from scipy.sparse import coo_matrix


class SparseBlockHessian:
    def __init__(self, hvp, sparsity_array):
        # sparse_hessian_lib.py:30-60
        self._hvp = hvp
        self._sparsity_array = np.asarray(sparsity_array)
        self._num_blocks = self._sparsity_array.shape[0]
        self._block_size = self._sparsity_array.shape[1]
        if len(np.unique(sparsity_array)) != len(self._sparsity_array.flatten()):
            raise ValueError('The indices in ``sparsity array`` must be unique.')

    def _hess_summed_term(self, opt_par, ib):
        # sparse_hessian_lib.py:62-67
        v = np.zeros_like(opt_par)
        v[self._sparsity_array[:, ib]] = 1
        return self._hvp(opt_par, v)

    def get_block_hessian(self, opt_par, print_every=0):
        # sparse_hessian_lib.py:69-108
        opt_par = np.atleast_1d(opt_par)
        if opt_par.ndim != 1:
            raise ValueError('``opt_par`` must be a vector.')
        mat_vals, mat_rows, mat_cols = [], [], []
        for ib in range(self._block_size):
            hess_prod = self._hess_summed_term(opt_par, ib)
            for b in range(self._num_blocks):
                hess_inds = self._sparsity_array[b, :]
                mat_vals.extend(hess_prod[hess_inds])
                mat_rows.extend(hess_inds)
                mat_cols.extend(np.full(self._block_size, hess_inds[ib]))
        d = len(opt_par)
        return coo_matrix((mat_vals, (mat_rows, mat_cols)), (d, d))

    def get_global_hessian(self, opt_par, global_inds=None, print_every=0):
        # sparse_hessian_lib.py:110-163
        local_inds = np.hstack(self._sparsity_array)
        if global_inds is None:
            global_inds = np.setdiff1d(np.arange(len(opt_par)), local_inds)
        inter = np.intersect1d(global_inds, local_inds)
        if len(inter) > 0:
            raise ValueError(
                'The global and local indices must be disjoint.  {}'.format(inter))
        mat_vals, mat_rows, mat_cols = [], [], []
        v = np.zeros_like(opt_par)
        for ig in global_inds:
            v[ig] = 1
            hess_row = self._hvp(opt_par, v)
            for il in local_inds:
                mat_vals.append(hess_row[il]); mat_cols.append(ig); mat_rows.append(il)
                mat_vals.append(hess_row[il]); mat_cols.append(il); mat_rows.append(ig)
            for ig2 in global_inds:
                mat_vals.append(0.5 * hess_row[ig2]); mat_cols.append(ig); mat_rows.append(ig2)
                mat_vals.append(0.5 * hess_row[ig2]); mat_cols.append(ig2); mat_rows.append(ig)
            v[ig] = 0
        d = len(opt_par)
        return coo_matrix((mat_vals, (mat_rows, mat_cols)), (d, d))

    def get_hessian(self, opt_par, print_every=0):
        # sparse_hessian_lib.py:165-168
        return self.get_block_hessian(opt_par) + self.get_global_hessian(opt_par)
