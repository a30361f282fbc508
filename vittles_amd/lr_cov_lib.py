"""Drop-in for ``vittles/lr_cov_lib.py`` (reference lines 10-216): linear-response covariances
``J1 H^{-1} J2^T`` with the solves on the MI355X.

Same constructor, methods, shape checks and error messages.  ``objective_fun`` must be a
registered objective (``grad(opt)``, and ``hessian(opt)`` when ``hessian_at_opt`` is not
given); ``calculate_moments`` arguments must expose ``jacobian(opt_par)`` -- the reference
obtains both with autograd, which is not part of this build.
"""
from copy import deepcopy

import numpy as np
import torch

from . import device as dev
from . import solver_lib, rows


class LinearResponseCovariances:
    def __init__(self, objective_fun, opt_par_value, validate_optimum=False,
                 hessian_at_opt=None, factorize_hessian=True, grad_tol=1e-8):
        # reference :43-86
        self._obj_fun = objective_fun
        self._grad_tol = grad_tol
        # a group-sharded registered objective (one rank per GPU): this rank's opt_par is [its local entries ...,
        # n_global replicated global entries]; sums over the parameter axis are all-reduced, globals counted once
        self._group = getattr(objective_fun, 'process_group', None)
        self._n_global = int(getattr(objective_fun, 'n_global', 0)) if self._group is not None else 0
        self.set_base_values(opt_par_value, hessian_at_opt, factorize_hessian, validate=validate_optimum)

    def set_base_values(self, opt_par_value, hessian_at_opt, factorize_hessian=True,
                        validate=True, grad_tol=None):
        # reference :88-119 (``factorize_hessian`` accepted and ignored, as at :106)
        if grad_tol is None:
            grad_tol = self._grad_tol
        self._opt0 = deepcopy(opt_par_value)

        if hessian_at_opt is None:
            if not hasattr(self._obj_fun, 'hessian'):
                raise NotImplementedError(
                    'No Hessian: pass ``hessian_at_opt`` or a registered objective '
                    '(autograd is not part of this build).')
            H = self._obj_fun.hessian(dev.to_device(self._opt0))
            self._hess0 = dev.like_input(H, self._opt0)
        else:
            self._hess0 = hessian_at_opt

        self.hess_solver = solver_lib.get_cholesky_solver(self._hess0)

        if validate:
            if not hasattr(self._obj_fun, 'grad'):
                raise NotImplementedError(
                    'validate_optimum needs the gradient: pass a registered objective '
                    '(autograd is not part of this build).')
            grad0 = self._obj_fun.grad(dev.to_device(self._opt0))
            newton_step = -1 * self.hess_solver(grad0)
            newton_step_norm = self._param_norm(dev.to_device(newton_step))
            if newton_step_norm > grad_tol:
                err_msg = \
                    'The gradient is not zero at the proposed optimal ' + \
                    'values.  ||newton_step|| = {} > {} = grad_tol'.format(
                        newton_step_norm, grad_tol)
                raise ValueError(err_msg)

    def _param_norm(self, vec):
        """2-norm over the WHOLE parameter vector: for a sharded objective the local entries of every rank plus
        the replicated globals once."""
        if self._group is None:
            return rows.norm(vec)
        from . import sharding
        nl = vec.shape[0] - self._n_global
        sq = rows.dot(vec[:nl], vec[:nl])
        sharding.all_reduce_sum(sq, self._group)
        return float((sq + rows.dot(vec[nl:], vec[nl:])).sqrt().item())

    def get_hessian_at_opt(self):
        return self._hess0

    def get_lr_covariance_from_jacobians(self, moment_jacobian1, moment_jacobian2):
        # reference :124-172
        if moment_jacobian1.ndim != 2:
            raise ValueError('moment_jacobian1 must be a 2d array.')

        if moment_jacobian2.ndim != 2:
            raise ValueError('moment_jacobian2 must be a 2d array.')

        if moment_jacobian1.shape[1] != len(self._opt0):
            err_msg = ('The number of rows of moment_jacobian1 must match' +
                       'the dimension of the optimization parameter. ' +
                       'Expected {} rows, but got shape = {}').format(
                         len(self._opt0), moment_jacobian1.shape)
            raise ValueError(err_msg)

        if moment_jacobian2.shape[1] != len(self._opt0):
            err_msg = ('The number of rows of moment_jacobian2 must match' +
                       'the dimension of the optimization parameter. ' +
                       'Expected {} rows, but got shape = {}').format(
                         len(self._opt0), moment_jacobian2.shape)
            raise ValueError(err_msg)

        # moment_jacobian1 @ hess_solver(moment_jacobian2.T), products on the device:
        # Z = H^{-1} J2^T is (d, K2); the result (J1^T)^T Z is ONE fp64-MFMA product over the d axis
        # (vt_gemm_tn; outside its shape limits: one streaming pass per moment).
        J1 = dev.to_device(moment_jacobian1)
        same = moment_jacobian2 is moment_jacobian1
        J2t = (J1 if same else dev.to_device(moment_jacobian2)).t().contiguous()       # (d, K2)
        Z = dev.to_device(self.hess_solver(J2t)).contiguous()
        K1, K2 = J1.shape[0], Z.shape[1]
        if self._group is not None:
            # sharded: every rank holds the columns of the Jacobians that belong to ITS parameters (+ the
            # replicated global columns); the sum over the parameter axis is all-reduced, globals counted once
            from . import sharding
            J1t = J1.t().contiguous()
            if sharding.group_rank(self._group) != 0 and self._n_global > 0:
                J1t[J1t.shape[0] - self._n_global:] = 0.0
            out = rows.gemm_tn_tiled(J1t, Z)
            sharding.all_reduce_sum(out, self._group)
        elif K1 >= 1 and K2 >= 1:
            J1t = J2t if same else J1.t().contiguous()                                   # (d, K1)
            out = rows.gemm_tn_tiled(J1t, Z)      # (more than 8192 moments: panels of 8192, still vt_gemm_tn)
        else:
            out = torch.empty((K1, K2), dtype=torch.float64, device=Z.device)
        return dev.like_input(out, moment_jacobian1)

    def get_moment_jacobian(self, calculate_moments):
        # reference :174-192 uses autograd.jacobian(calculate_moments)
        if not hasattr(calculate_moments, 'jacobian'):
            raise NotImplementedError(
                '``calculate_moments`` must expose ``jacobian(opt_par)`` '
                '(autograd is not part of this build).')
        return calculate_moments.jacobian(self._opt0)

    def get_lr_covariance(self, calculate_moments):
        # reference :194-216
        moment_jacobian = self.get_moment_jacobian(calculate_moments)
        return self.get_lr_covariance_from_jacobians(moment_jacobian, moment_jacobian)


class LinearMoments:
    """``calculate_moments(opt_par) = A @ opt_par`` with its (constant) Jacobian -- e.g. the
    variational means of a mean-field Gaussian (tests/test_lr_cov_lib.py:86-90)."""

    def __init__(self, A):
        self.A = A

    def __call__(self, opt_par):
        return self.A @ opt_par

    def jacobian(self, opt_par):
        return self.A
