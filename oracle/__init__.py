"""CPU oracle for the vittles linear-response sensitivity hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``vittles_amd/`` may import this
package.  The only permitted users are ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``, and there only as the checker
(or the timed CPU baseline), never as a product code path.

What it is: a NumPy/SciPy restatement of the reference algorithm
(``/root/reference/vittles``), each function citing the reference file:line it
follows.  The reference differentiates a user callable with ``autograd`` (absent
from this image and un-vendored; ``requirements.txt:1`` pins only
``autograd>=1.3``), so the derivative pieces are restated in closed form for the
registered objectives and the linear-algebra pieces call the very same
LAPACK/SuperLU entry points the reference calls through SciPy.

Pinning status (see DESIGN.md "Oracle"):
  * solver_lib dense / sparse closures: pinned against the reference's own
    ``solver_lib.py`` run in the build container (``tests/golden/make_golden.py``)
    and against ``tests/test_solver_lib.py:11-43`` (seed 101 matrix).
  * linear approximation: pinned against the reference tests' closed-form
    known answers (QuadraticModel ``tests/test_utils.py:23-75``; weighted linear
    regression ``tests/test_sensitivity_lib.py:838-901``), against the printed
    outputs of the reference's own notebook
    (``docs/source/example_notebooks/mle_weight_sensitivity_example.ipynb``
    cells 16, 24) and against an independent fp64 AD engine (``torch.func``).
  * LR covariance: pinned against ``tests/test_lr_cov_lib.py:16-135``.
  * block Hessian: structure pinned against ``tests/test_sparse_hessian_lib.py``
    (dense == block + global); values need paragami -> restated with fixed
    matrices.
  * CG: the reference's ``get_cg_solver`` cannot run on SciPy >= 1.14
    (``atol='legacy'`` is rejected), so the CG iteration is a restatement of the
    published SciPy <= 1.11 algorithm: "parity unpinned" beyond agreement with
    ``np.linalg.solve`` on the reference's test matrix.
  * weighted-logistic / GLM / hierarchical-VI numbers at 1e-10: no reference
    test pins them; pinned by closed form == torch.func AD only.
"""
