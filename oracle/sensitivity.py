"""ORACLE (test infrastructure, never a product path): restatement of the linear
approximation classes of ``/root/reference/vittles/sensitivity_lib.py`` with the
autograd pieces replaced by explicit derivative callables.

  ``grad(theta, w) -> (D,)``, ``hess(theta, w) -> (D, D)``,
  ``cross_hess(theta, w) -> (D, N)``  stand in for
  ``autograd.grad`` (:354), ``autograd.hessian`` (:382) and
  ``autograd.jacobian(grad, argnum=1)`` (:180-181).
"""
from copy import deepcopy

import numpy as np

from . import solvers


class EstimatingEquationLinearApproximation:
    # sensitivity_lib.py:85-254
    def __init__(self, estimating_equation, input_val0, hyper_val0, hess_solver,
                 validate_solution=False, estimating_equation_jac0=None,
                 hyper_par_estimating_equation_jac=None, solution_tol=1e-8):
        self._estimating_equation = estimating_equation
        self._hyper_ee_fun_jac = hyper_par_estimating_equation_jac
        self._hess_solver = hess_solver
        self._solution_tol = solution_tol
        self.set_location(input_val0, hyper_val0, estimating_equation_jac0,
                          validate_solution=validate_solution, solution_tol=solution_tol)

    def set_location(self, input_val0, hyper_val0, estimating_equation_jac0,
                     validate_solution=True, solution_tol=None):
        # sensitivity_lib.py:192-226 (order of checks, deep copies, sign)
        self._input_val0 = deepcopy(input_val0)
        self._hyper_val0 = deepcopy(hyper_val0)
        if validate_solution:
            if solution_tol is None:
                solution_tol = self._solution_tol
            ee_val = self._estimating_equation(self._input_val0, self._hyper_val0)
            ee_norm = np.linalg.norm(ee_val)
            if ee_norm > solution_tol:
                raise ValueError(
                    'The estimating equation is not zero at the proposed  ' +
                    'values.  ||ee|| = {} > {} = solution_tol'.format(ee_norm, solution_tol))
        if estimating_equation_jac0 is None:
            self._estimating_equation_jac0 = self._hyper_ee_fun_jac(self._input_val0, self._hyper_val0)
        else:
            self._estimating_equation_jac0 = estimating_equation_jac0
        if self._estimating_equation_jac0.shape != (len(self._input_val0), len(self._hyper_val0)):
            raise ValueError('``_estimating_equation_jac0`` is the wrong shape.')
        self._sens_mat = -1 * self._hess_solver(self._estimating_equation_jac0)

    def get_dinput_dhyper(self):
        return self._sens_mat

    def hess_solver(self):
        return self._hess_solver

    def predict_input_par_from_hyper_par(self, new_hyper_par_value):
        # sensitivity_lib.py:236-247
        return self._input_val0 + self._sens_mat @ (new_hyper_par_value - self._hyper_val0)


class HyperparameterSensitivityLinearApproximation(EstimatingEquationLinearApproximation):
    # sensitivity_lib.py:258-429
    def __init__(self, grad, hess, cross_hess, opt_par_value, hyper_par_value,
                 validate_optimum=False, hessian_at_opt=None, cross_hess_at_opt=None,
                 grad_tol=1e-8):
        self._hess_fun = hess
        hess_solver = self._get_hessian_solver(opt_par_value, hyper_par_value, hessian_at_opt)
        EstimatingEquationLinearApproximation.__init__(
            self, estimating_equation=grad, input_val0=opt_par_value, hyper_val0=hyper_par_value,
            hess_solver=hess_solver, validate_solution=validate_optimum,
            estimating_equation_jac0=cross_hess_at_opt,
            hyper_par_estimating_equation_jac=cross_hess, solution_tol=grad_tol)

    def _get_hessian_solver(self, opt_par_value, hyper_par_value, hessian_at_opt):
        # sensitivity_lib.py:376-390
        if hessian_at_opt is None:
            self._hess0 = self._hess_fun(opt_par_value, hyper_par_value)
        else:
            self._hess0 = hessian_at_opt
        if self._hess0.shape != (len(opt_par_value), len(opt_par_value)):
            raise ValueError('``hessian_at_opt`` is the wrong shape.')
        return solvers.get_cholesky_solver(self._hess0)

    def get_dopt_dhyper(self):
        return self.get_dinput_dhyper()

    def get_hessian_at_opt(self):
        return self._hess0

    def predict_opt_par_from_hyper_par(self, new_hyper_par_value):
        return self.predict_input_par_from_hyper_par(new_hyper_par_value)
