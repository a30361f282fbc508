"""CPU tests (no GPU) of the host logic that recovers block-arrowhead structure from a scipy sparse matrix
(vittles_amd.sparse_hessian_lib.detect_arrowhead): what lets get_sparse_cholesky_solver take ANY sparse matrix of
that form, as the reference's SuperLU route does (vittles/solver_lib.py:33-48)."""
import numpy as np
import pytest
import scipy.sparse as sp


def _spd(rng, n):
    q = rng.standard_normal((n, n))
    return q @ q.T + n * np.eye(n)


def test_uniform_blocks_permuted_with_dense_border():
    from vittles_amd import sparse_hessian_lib as shl
    rng = np.random.default_rng(0)
    G, M, Kg = 300, 5, 3
    d = G * M + Kg
    A = np.zeros((d, d))
    for g in range(G):
        A[g * M:(g + 1) * M, g * M:(g + 1) * M] = _spd(rng, M)
    A[:G * M, G * M:] = rng.standard_normal((G * M, Kg))
    A[G * M:, :G * M] = A[:G * M, G * M:].T
    A[G * M:, G * M:] = _spd(rng, Kg) + G
    perm = rng.permutation(d)
    Ap = A[np.ix_(perm, perm)]
    inds, ginds, d_pad = shl.detect_arrowhead(sp.csc_matrix(Ap))
    assert inds.shape == (G, M) and d_pad == d
    inv = np.argsort(perm)                       # original index i sits at position inv[i]
    assert sorted(ginds) == sorted(inv[G * M:])
    orig = {frozenset(inv[g * M:(g + 1) * M]) for g in range(G)}
    assert all(frozenset(row) in orig for row in inds)
    assert shl.matrix_fingerprint(sp.csc_matrix(Ap)) == shl.matrix_fingerprint(sp.csc_matrix(Ap).copy())


def test_ragged_components_are_packed_and_padded():
    from vittles_amd import sparse_hessian_lib as shl
    rng = np.random.default_rng(1)
    sizes = [5, 3, 1, 5, 2]
    d = sum(sizes)
    A = np.zeros((d, d))
    o = 0
    for s in sizes:
        A[o:o + s, o:o + s] = _spd(rng, s)
        o += s
    inds, ginds, d_pad = shl.detect_arrowhead(sp.csr_matrix(A))
    assert len(ginds) == 0 and inds.shape[1] == 5
    flat = inds.reshape(-1)
    assert len(np.unique(flat)) == flat.size                       # every index once
    assert set(flat[flat < d]) == set(range(d))                    # all real unknowns covered
    assert d_pad == inds.size and set(flat[flat >= d]) == set(range(d, d_pad))   # the rest is padding
    # no bin mixes entries that are coupled with another bin
    comp_of = np.repeat(np.arange(len(sizes)), sizes)
    for row in inds:
        real = row[row < d]
        for c in set(comp_of[real]):
            assert set(np.nonzero(comp_of == c)[0]) <= set(real)


def test_low_degree_global_is_found_by_component_size():
    """A coupling variable that touches fewer rows than the degree threshold still merges its blocks into one
    oversized component: its best-connected node is promoted to a global."""
    from vittles_amd import sparse_hessian_lib as shl
    rng = np.random.default_rng(2)
    G, M = 40, 3                                  # 120 local unknowns + 1 global of degree 121 < 2 * MAX_BLOCK
    d = G * M + 1
    A = np.zeros((d, d))
    for g in range(G):
        A[g * M:(g + 1) * M, g * M:(g + 1) * M] = _spd(rng, M)
    A[:-1, -1] = A[-1, :-1] = 0.1
    A[-1, -1] = 50.0
    inds, ginds, d_pad = shl.detect_arrowhead(sp.coo_matrix(A))
    assert list(ginds) == [d - 1] and inds.shape == (G, M) and d_pad == d


def test_not_arrowhead_and_explicit_keywords():
    from vittles_amd import sparse_hessian_lib as shl
    n = 500
    T = sp.diags([np.full(n - 1, -1.0), np.full(n, 4.0), np.full(n - 1, -1.0)], [-1, 0, 1]).tocsc()
    assert shl.detect_arrowhead(T) is None        # a chain: one component, no small set of globals splits it
    sa = np.arange(12).reshape(4, 3)
    inds, ginds, d_pad = shl.detect_arrowhead(sp.identity(14, format="csr"), sparsity_array=sa)
    assert np.array_equal(inds, sa) and list(ginds) == [12, 13] and d_pad == 14
    with pytest.raises(ValueError, match="must be unique"):
        shl.detect_arrowhead(sp.identity(4, format="csr"), sparsity_array=np.array([[0, 1], [1, 2]]))
    with pytest.raises(ValueError, match="must be disjoint"):
        shl.detect_arrowhead(sp.identity(4, format="csr"), sparsity_array=np.array([[0, 1]]), global_inds=[1, 2])
