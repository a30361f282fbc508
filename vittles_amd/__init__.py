"""vittles_amd: the linear-response sensitivity hot path of rgiordan/vittles on MI355X.

Public names mirror ``vittles/__init__.py:1-8`` of the reference.
"""
from .sensitivity_lib import HyperparameterSensitivityLinearApproximation
from .sensitivity_lib import EstimatingEquationLinearApproximation
from .objectives import RegisteredObjective, WeightedGLMObjective
from . import solver_lib

__all__ = [
    "HyperparameterSensitivityLinearApproximation",
    "EstimatingEquationLinearApproximation",
    "RegisteredObjective",
    "WeightedGLMObjective",
    "solver_lib",
]
