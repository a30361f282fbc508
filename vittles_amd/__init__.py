"""vittles_amd: the linear-response sensitivity hot path of rgiordan/vittles on MI355X.

Public names mirror ``vittles/__init__.py:1-8`` of the reference.
"""
from .sensitivity_lib import HyperparameterSensitivityLinearApproximation
from .sensitivity_lib import EstimatingEquationLinearApproximation
from .sensitivity_lib import ParametricSensitivityTaylorExpansion
from .objectives import RegisteredObjective, WeightedGLMObjective, HierGaussMFVIObjective
from .sparse_hessian_lib import SparseBlockHessian
from .lr_cov_lib import LinearResponseCovariances
from . import solver_lib
from . import optim
from . import synth
from . import bivariate_sensitivity_lib


__version__ = "0.1.0"

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
