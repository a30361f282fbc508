"""Host side of the counter-based generator (vittles_amd/synth.py): Philox-4x32-10 known answers and the
statistics of the NumPy mirror.  The device kernels are pinned to this mirror in tests/test_gpu_synth.py."""
import numpy as np


def test_philox_known_answer():
    """Random123's published known-answer vectors for philox4x32-10 (kat_vectors: zero counter / key, and the
    all-ones case)."""
    from vittles_amd import synth
    out = [int(v[0]) for v in synth.philox4x32_10([0], [0], [0], [0], 0)]
    assert out == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    out = [int(v[0]) for v in synth.philox4x32_10([f], [f], [f], [f], (f << 32) | f)]
    assert out == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_host_rows_are_position_independent():
    from vittles_amd import synth
    X = synth.normal_rows_host(5, 0, 300, 7)
    assert np.array_equal(synth.normal_rows_host(5, 100, 50, 7), X[100:150])
    assert np.array_equal(synth.normal_rows_host(5, 0, 0, 7, row_index=[299, 0, 17]), X[[299, 0, 17]])
    assert abs(X.mean()) < 0.1 and abs(X.std() - 1.0) < 0.1
    z = X @ synth.theta_star_host(5, 7)
    y = synth.response_host(5, z, "logistic")
    assert set(np.unique(y)) <= {0.0, 1.0}
    assert np.array_equal(synth.response_host(5, z[40:60], "logistic", row0=40), y[40:60])
    k = synth.response_host(5, 0.5 * z, "poisson")
    assert np.all(k >= 0) and np.all(k == np.floor(k))
