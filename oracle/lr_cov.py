"""ORACLE (test infrastructure, never a product path): restatement of
``/root/reference/vittles/lr_cov_lib.py`` with the autograd pieces replaced by
explicit callables (``grad(opt) -> (d,)``, ``hess(opt) -> (d, d)``) and moment
Jacobians handed in as arrays (``autograd.jacobian(calculate_moments)``,
lr_cov_lib.py:191-192).
"""
from copy import deepcopy

import numpy as np

from . import solvers


class LinearResponseCovariances:
    def __init__(self, grad, hess, opt_par_value, validate_optimum=False,
                 hessian_at_opt=None, factorize_hessian=True, grad_tol=1e-8):
        # lr_cov_lib.py:43-86
        self._obj_fun_grad = grad
        self._obj_fun_hessian = hess
        self._grad_tol = grad_tol
        self.set_base_values(opt_par_value, hessian_at_opt, factorize_hessian,
                             validate=validate_optimum)

    def set_base_values(self, opt_par_value, hessian_at_opt, factorize_hessian=True,
                        validate=True, grad_tol=None):
        # lr_cov_lib.py:88-119 (``factorize_hessian`` is accepted and ignored, :106)
        if grad_tol is None:
            grad_tol = self._grad_tol
        self._opt0 = deepcopy(opt_par_value)
        if hessian_at_opt is None:
            self._hess0 = self._obj_fun_hessian(self._opt0)
        else:
            self._hess0 = hessian_at_opt
        self.hess_solver = solvers.get_cholesky_solver(self._hess0)
        if validate:
            grad0 = self._obj_fun_grad(self._opt0)
            newton_step = -1 * self.hess_solver(grad0)
            newton_step_norm = np.linalg.norm(newton_step)
            if newton_step_norm > grad_tol:
                raise ValueError(
                    'The gradient is not zero at the proposed optimal ' +
                    'values.  ||newton_step|| = {} > {} = grad_tol'.format(
                        newton_step_norm, grad_tol))

    def get_hessian_at_opt(self):
        return self._hess0

    def get_lr_covariance_from_jacobians(self, moment_jacobian1, moment_jacobian2):
        # lr_cov_lib.py:124-172
        if moment_jacobian1.ndim != 2:
            raise ValueError('moment_jacobian1 must be a 2d array.')
        if moment_jacobian2.ndim != 2:
            raise ValueError('moment_jacobian2 must be a 2d array.')
        if moment_jacobian1.shape[1] != len(self._opt0):
            raise ValueError(('The number of rows of moment_jacobian1 must match' +
                              'the dimension of the optimization parameter. ' +
                              'Expected {} rows, but got shape = {}').format(
                                  len(self._opt0), moment_jacobian1.shape))
        if moment_jacobian2.shape[1] != len(self._opt0):
            raise ValueError(('The number of rows of moment_jacobian2 must match' +
                              'the dimension of the optimization parameter. ' +
                              'Expected {} rows, but got shape = {}').format(
                                  len(self._opt0), moment_jacobian2.shape))
        return moment_jacobian1 @ self.hess_solver(moment_jacobian2.T)
