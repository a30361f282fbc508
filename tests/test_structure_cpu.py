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


def _arrow(rng, G=40, M=4, Kg=2):
    d = G * M + Kg
    A = np.zeros((d, d))
    for g in range(G):
        A[g * M:(g + 1) * M, g * M:(g + 1) * M] = _spd(rng, M)
    A[:G * M, G * M:] = rng.standard_normal((G * M, Kg))
    A[G * M:, :G * M] = A[:G * M, G * M:].T
    A[G * M:, G * M:] = _spd(rng, Kg) + G
    return A


def test_one_triangle_or_non_symmetric_input_is_refused():
    """the structured solve reads (local row, global col) entries only; an input that stores one
    triangle, or is not symmetric, must raise instead of silently solving another matrix (the reference's LU,
    vittles/solver_lib.py:48, would have taken it)."""
    from vittles_amd import sparse_hessian_lib as shl
    rng = np.random.default_rng(3)
    A = _arrow(rng)
    d = A.shape[0]
    inds = np.arange(40 * 4).reshape(40, 4)
    with pytest.raises(ValueError, match="SYMMETRIC"):
        shl.arrowhead_from_sparse(sp.csc_matrix(np.triu(A)), sparsity_array=inds)
    with pytest.raises(ValueError, match="SYMMETRIC"):
        shl.arrowhead_from_sparse(sp.csc_matrix(np.tril(A)), sparsity_array=inds)
    B = A.copy()
    B[3, d - 1] *= 1.001                         # border entry without its mirror image
    with pytest.raises(ValueError, match="border"):
        shl.arrowhead_from_sparse(sp.coo_matrix(B), sparsity_array=inds)
    C = A.copy()
    C[1, 2] += 0.5                               # inside a block
    with pytest.raises(ValueError, match="block diagonal"):
        shl.arrowhead_from_sparse(sp.coo_matrix(C), sparsity_array=inds)
    E = A.copy()
    E[d - 1, d - 2] += 0.5                       # inside the corner
    with pytest.raises(ValueError, match="corner"):
        shl.arrowhead_from_sparse(sp.coo_matrix(E), sparsity_array=inds)


def test_fingerprint_sees_swapped_and_moved_entries():
    """in-place edits that keep nnz and the sums must still change the fingerprint."""
    from vittles_amd import sparse_hessian_lib as shl
    rng = np.random.default_rng(4)
    h = sp.coo_matrix(_arrow(rng))
    f0 = shl.matrix_fingerprint(h)
    assert f0 == shl.matrix_fingerprint(h.copy())
    g = h.copy()
    g.data[[0, 5]] = g.data[[5, 0]]              # swapped values: same sum, same abs-sum
    assert shl.matrix_fingerprint(g) != f0
    g = h.copy()
    g.row[[0, 700]] = g.row[[700, 0]]            # moved entries: values untouched
    assert shl.matrix_fingerprint(g) != f0
    c = h.tocsc()
    fc = shl.matrix_fingerprint(c)
    c.indices[:2] = c.indices[:2][::-1].copy()
    assert shl.matrix_fingerprint(c) != fc
