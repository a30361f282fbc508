"""vittles_amd: the linear-response sensitivity hot path of rgiordan/vittles on MI355X.

Public names mirror ``vittles/__init__.py:1-8`` of the reference.
"""
from .sensitivity_lib import HyperparameterSensitivityLinearApproximation
from .sensitivity_lib import EstimatingEquationLinearApproximation
from .objectives import RegisteredObjective, WeightedGLMObjective, HierGaussMFVIObjective
from .sparse_hessian_lib import SparseBlockHessian
from .lr_cov_lib import LinearResponseCovariances
from . import solver_lib
from . import optim
from . import bivariate_sensitivity_lib


class ParametricSensitivityTaylorExpansion:
    """Name parity with vittles/__init__.py:4.  Higher-order Taylor expansions need
    arbitrary-order automatic differentiation (sensitivity_lib.py:1021-1363, "experimental"
    per its own docstring) and are outside this build's hot path."""

    def __init__(self, *args, **kwargs):
        raise NotImplementedError('ParametricSensitivityTaylorExpansion is outside the MI355X hot path.')

__all__ = [
    "HyperparameterSensitivityLinearApproximation",
    "EstimatingEquationLinearApproximation",
    "RegisteredObjective",
    "WeightedGLMObjective",
    "HierGaussMFVIObjective",
    "SparseBlockHessian",
    "LinearResponseCovariances",
    "ParametricSensitivityTaylorExpansion",
    "solver_lib",
    "optim",
    "bivariate_sensitivity_lib",
]
