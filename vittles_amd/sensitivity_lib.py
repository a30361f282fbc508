"""Drop-in for the linear-approximation classes of ``vittles/sensitivity_lib.py``
(reference: /root/reference/vittles/sensitivity_lib.py:85-429), computing on the
MI355X.  Same constructor arguments, same methods, same error messages, same
sign / shape conventions:  ``S = -H^{-1} jac0`` with shape (len(opt), len(hyper)).

Differences that the absence of autograd forces:
  * ``objective_fun`` must be a ``RegisteredObjective`` (closed-form derivatives in
    HIP kernels) unless BOTH ``hessian_at_opt`` and ``cross_hess_at_opt`` are given
    and ``validate_optimum`` is False -- then any callable is accepted, exactly the
    inputs the reference would not differentiate either.
  * Results have the container type of ``opt_par_value``: numpy in -> numpy out
    (``get_dopt_dhyper()`` is then F-contiguous (D, N) as SciPy's cho_solve returns
    it), CUDA tensor in -> CUDA tensors out (a transposed view of the row-major
    (N, D) device buffer: no copy, stays sharded like X).
"""
from copy import deepcopy

import numpy as np
import torch

from . import device as dev
from . import sharding, solver_lib
from .objectives import RegisteredObjective, LazyCrossHessian


def get_linear_function(return_val0, arg_val0, dreturn_darg_rows, process_group=None):
    """Reference: ``get_linear_function`` (sensitivity_lib.py:18-82): a differentiable function
    that returns ``return_val0`` when evaluated at ``arg_val0`` and whose Jacobian is the
    pre-computed ``dreturn_darg``.  The reference registers an autograd primitive with
    ``vjp(g) = g.T @ dreturn_darg`` (:60-68) and ``jvp(g) = dreturn_darg @ g`` (:73-76); this is
    the same thing as a ``torch.autograd.Function`` (reverse AND forward mode), both products
    being one streaming pass over the device-resident Jacobian.

    ``dreturn_darg_rows``: the Jacobian TRANSPOSED, (len(arg), len(return)) row-major on the
    device (how the sensitivity matrix is stored).  ``process_group``: the argument is ROW-SHARDED
    over the ranks of that group (each rank holds its slice of ``arg`` and the matching rows of the
    Jacobian): the JVP ``J t = sum_ranks J_r t_r`` is all-reduced, the VJP ``g^T J_r`` is this
    rank's slice of the gradient and needs no communication.  As in the reference the function must be
    evaluated at ``arg_val0`` (ValueError otherwise) and is only once differentiable."""
    from torch.autograd.function import once_differentiable
    from . import rows
    devc = dreturn_darg_rows.device
    ret0 = dev.to_device(return_val0, devc)
    arg0 = dev.to_device(arg_val0, devc)

    def _check_arg_par_value(arg_par, tolerance=1e-8):
        if arg_par.shape != arg0.shape or float((arg_par - arg0).abs().max().item()) > tolerance:
            raise ValueError('get_return_par must be evaluated at ', arg_val0, ' != ', arg_par)

    class _ReturnPar(torch.autograd.Function):
        @staticmethod
        def forward(arg_par):
            return ret0.clone()

        @staticmethod
        def setup_context(ctx, inputs, output):
            pass

        @staticmethod
        @once_differentiable
        def backward(ctx, g):
            return rows.gemv(dreturn_darg_rows, g.to(torch.float64).contiguous())     # g.T @ J

        @staticmethod
        def jvp(ctx, t):
            out = rows.gemv_t(dreturn_darg_rows, t.to(torch.float64).contiguous())    # J @ t
            return sharding.all_reduce_sum(out, process_group)

    def get_return_par(arg_par):
        if isinstance(arg_par, torch.Tensor):
            _check_arg_par_value(arg_par.detach().to(devc))
            return _ReturnPar.apply(arg_par.to(devc))
        _check_arg_par_value(dev.to_device(arg_par, devc))
        return dev.like_input(ret0.clone(), arg_par)

    return get_return_par


def _len(x):
    return x.shape[0] if hasattr(x, "shape") else len(x)


def _norm(x):
    if isinstance(x, torch.Tensor):
        # fixed-order device reduction (vt_dot3) whatever the tensor's device or dtype was
        from . import rows
        return rows.norm(dev.to_device(x.detach().to(torch.float64).reshape(-1)))
    return float(np.linalg.norm(x))


class EstimatingEquationLinearApproximation:
    """Reference: sensitivity_lib.py:85-254."""

    def __init__(self, estimating_equation, input_val0, hyper_val0, hess_solver,
                 validate_solution=False, estimating_equation_jac0=None,
                 hyper_par_estimating_equation=None, solution_tol=1e-8):
        self._estimating_equation = estimating_equation
        if hyper_par_estimating_equation is None:
            self._hyper_par_estimating_equation = self._estimating_equation
        else:
            self._hyper_par_estimating_equation = hyper_par_estimating_equation
        # The reference builds autograd.jacobian(hyper_ee, argnum=1) here (:180-181);
        # a registered estimating equation carries that Jacobian itself.
        self._hyper_ee_fun_jac = getattr(self._hyper_par_estimating_equation, "jacobian_hyper", None)
        self._hess_solver = hess_solver
        self._solution_tol = solution_tol
        # a row-sharded registered objective (one rank per GPU, each holding a slice of the hyper-
        # parameter = data weights): products that sum over the hyper axis are all-reduced
        if not hasattr(self, '_process_group'):
            self._process_group = getattr(getattr(estimating_equation, '_obj', None), 'process_group', None)
        self.set_location(input_val0, hyper_val0, estimating_equation_jac0,
                          validate_solution=validate_solution, solution_tol=solution_tol)

    def set_location(self, input_val0, hyper_val0, estimating_equation_jac0,
                     validate_solution=True, solution_tol=None):
        # Reference: sensitivity_lib.py:192-226 (same order: copies, check, jac0, shape, solve).
        self._input_val0 = deepcopy(input_val0)
        self._hyper_val0 = deepcopy(hyper_val0)
        self._as_torch = isinstance(input_val0, torch.Tensor)

        if validate_solution:
            if solution_tol is None:
                solution_tol = self._solution_tol
            if self._estimating_equation is None:
                raise NotImplementedError(
                    'validate_optimum needs the gradient of the objective: pass a RegisteredObjective '
                    '(autograd is not part of this build).')
            ee_val = self._estimating_equation(self._input_val0, self._hyper_val0)
            ee_norm = _norm(ee_val)
            if ee_norm > solution_tol:
                err_msg = \
                    'The estimating equation is not zero at the proposed  ' + \
                    'values.  ||ee|| = {} > {} = solution_tol'.format(
                        ee_norm, solution_tol)
                raise ValueError(err_msg)

        if estimating_equation_jac0 is None:
            if self._hyper_ee_fun_jac is None:
                raise NotImplementedError(
                    'No cross-Hessian: pass ``cross_hess_at_opt`` or a RegisteredObjective '
                    '(autograd is not part of this build).')
            self._estimating_equation_jac0 = self._hyper_ee_fun_jac(self._input_val0, self._hyper_val0)
        else:
            self._estimating_equation_jac0 = estimating_equation_jac0
        if tuple(self._estimating_equation_jac0.shape) != \
                (_len(self._input_val0), _len(self._hyper_val0)):
            raise ValueError('``_estimating_equation_jac0`` is the wrong shape.')

        jac0 = self._estimating_equation_jac0
        chol = getattr(self._hess_solver, "cholesky", None)
        self._sens_rows = None      # (N, D) row-major CUDA tensor when computed on the fast path
        self._sens_host = None
        if isinstance(jac0, LazyCrossHessian):
            if chol is not None and chol.D <= 8192:
                # K2 + K5 fused: S[n, :] = -r_n H^{-1} x_n, jac0 never materialised.
                self._sens_rows = chol.solve_rows(jac0.X, rowscale=jac0.r, alpha=-1.0)
            else:
                self._sens_rows = (-1 * self._hess_solver(jac0.materialize())).t().contiguous()
        elif chol is not None and chol.D <= 8192 and _len(self._hyper_val0) >= 64:
            rows = dev.to_device(jac0, chol.L.device).t().contiguous()
            self._sens_rows = chol.solve_rows(rows, alpha=-1.0)
        else:
            sens = -1 * self._hess_solver(jac0)
            self._sens_rows = dev.to_device(sens).t().contiguous()

    @property
    def _sens_mat(self):
        """(D, N) sensitivity matrix in the caller's container type."""
        if self._as_torch:
            return self._sens_rows.t()
        if self._sens_host is None:
            self._sens_host = self._sens_rows.cpu().numpy().T   # F-contiguous (D, N)
        return self._sens_host

    def get_dinput_dhyper(self):
        return self._sens_mat

    def hess_solver(self):
        return self._hess_solver

    def predict_input_par_from_hyper_par(self, new_hyper_par_value):
        """``input_val0 + sens_mat @ (new_hyper - hyper_val0)`` (reference :236-247),
        as one streaming pass over the device-resident S."""
        from . import rows
        devc = self._sens_rows.device
        delta = dev.to_device(new_hyper_par_value, devc) - dev.to_device(self._hyper_val0, devc)
        if delta.ndim != 1 or delta.shape[0] != self._sens_rows.shape[0]:
            raise ValueError('new_hyper_par_value has the wrong shape.')
        part = rows.gemv_t(self._sens_rows, delta)          # this rank's rows of S: a partial sum
        sharding.all_reduce_sum(part, self._process_group)
        out = part + dev.to_device(self._input_val0, devc)
        return dev.like_input(out, self._input_val0)

    def predict_input_par_from_hyper_par_batch(self, new_hyper_par_values):
        """Build-side extension: the reference's
        ``predict_input_par_from_hyper_par`` (:236-247) for MANY hyper-parameter vectors at
        once.  ``new_hyper_par_values`` is (len(hyper), K): column k is one new hyper
        vector (a bootstrap draw, a K-fold or leave-one-out weight vector -- the notebook
        loops ``predict_opt_par_from_hyper_par`` over them, cells 24-26).  Returns
        (len(input), K): column k == ``predict_input_par_from_hyper_par(column k)``.
        One fp64-MFMA product ``W^T S`` over the device-resident S; ``W - hyper0`` is never
        formed: out = input0 - S^T hyper0 + S^T W."""
        from . import rows
        devc = self._sens_rows.device
        W = dev.to_device(new_hyper_par_values, devc)
        if W.ndim != 2 or W.shape[0] != self._sens_rows.shape[0]:
            raise ValueError('new_hyper_par_values must have shape (len(hyper_par), K).')
        sw0 = rows.gemv_t(self._sens_rows, dev.to_device(self._hyper_val0, devc))
        if self._process_group is None:
            bias = dev.to_device(self._input_val0, devc) - sw0
            out = rows.gemm_tn(W, self._sens_rows, bias=bias)        # (K, D) row-major
        else:
            # row-sharded: S^T W - S^T w0 summed over the ranks' rows, input0 added once
            out = rows.gemm_tn(W, self._sens_rows, bias=-sw0)
            sharding.all_reduce_sum(out, self._process_group)
            out += dev.to_device(self._input_val0, devc)
        return dev.like_input(out.t(), self._input_val0)

    def get_input_par_function(self):
        """Reference :249-254: the optimum as a differentiable function of the hyper-parameter
        (a ``torch.autograd.Function`` here instead of an autograd primitive)."""
        return get_linear_function(self._input_val0, self._hyper_val0, self._sens_rows,
                                   process_group=self._process_group)


class HyperparameterSensitivityLinearApproximation(EstimatingEquationLinearApproximation):
    """Reference: sensitivity_lib.py:258-429."""

    def __init__(self, objective_fun, opt_par_value, hyper_par_value,
                 validate_optimum=False, hessian_at_opt=None, cross_hess_at_opt=None,
                 hyper_par_objective_fun=None, grad_tol=1e-8):
        self._objective_fun = objective_fun
        self._process_group = getattr(objective_fun, 'process_group', None)
        registered = isinstance(objective_fun, RegisteredObjective)
        if not registered and not callable(objective_fun):
            raise TypeError('objective_fun must be callable.')
        if not registered and (hessian_at_opt is None or cross_hess_at_opt is None):
            raise NotImplementedError(
                'objective_fun is not a RegisteredObjective: pass both ``hessian_at_opt`` and '
                '``cross_hess_at_opt`` (autograd is not part of this build).')
        obj_fun_grad = _GradOf(objective_fun) if registered else None

        if hyper_par_objective_fun is None:
            hyper_par_objective_fun = self._objective_fun
        hyper_obj_fun_grad = _GradOf(hyper_par_objective_fun) \
            if isinstance(hyper_par_objective_fun, RegisteredObjective) else None

        hess_solver = self._get_hessian_solver(opt_par_value, hyper_par_value, hessian_at_opt)

        EstimatingEquationLinearApproximation.__init__(
            self,
            estimating_equation=obj_fun_grad,
            input_val0=opt_par_value,
            hyper_val0=hyper_par_value,
            hess_solver=hess_solver,
            validate_solution=validate_optimum,
            estimating_equation_jac0=cross_hess_at_opt,
            hyper_par_estimating_equation=hyper_obj_fun_grad,
            solution_tol=grad_tol)

    def _get_hessian_solver(self, opt_par_value, hyper_par_value, hessian_at_opt):
        # Reference: sensitivity_lib.py:376-390.
        if hessian_at_opt is None:
            self._hess0_dev = self._objective_fun.hessian(
                dev.to_device(opt_par_value), dev.to_device(hyper_par_value))
            self._hess0 = dev.like_input(self._hess0_dev, opt_par_value)
        else:
            self._hess0 = hessian_at_opt
        if tuple(self._hess0.shape) != (_len(opt_par_value), _len(opt_par_value)):
            raise ValueError('``hessian_at_opt`` is the wrong shape.')
        return solver_lib.get_cholesky_solver(self._hess0)

    def set_base_values(self, opt_par_value, hyper_par_value,
                        hessian_at_opt, cross_hess_at_opt,
                        validate_optimum=True, grad_tol=None):
        # Reference: sensitivity_lib.py:392-405 computes a new solver and then drops it
        # (the stale-solver bug noted in SURVEY.md section 5); here the new solver is used.
        self._hess_solver = self._get_hessian_solver(opt_par_value, hyper_par_value, hessian_at_opt)
        EstimatingEquationLinearApproximation.set_location(
            self, input_val0=opt_par_value, hyper_val0=hyper_par_value,
            estimating_equation_jac0=cross_hess_at_opt,
            validate_solution=validate_optimum, solution_tol=grad_tol)

    def get_dopt_dhyper(self):
        return super().get_dinput_dhyper()

    def get_hessian_at_opt(self):
        return self._hess0

    def predict_opt_par_from_hyper_par(self, new_hyper_par_value):
        return super().predict_input_par_from_hyper_par(new_hyper_par_value)

    def predict_opt_par_from_hyper_par_batch(self, new_hyper_par_values):
        return super().predict_input_par_from_hyper_par_batch(new_hyper_par_values)

    def get_opt_par_function(self):
        return super().get_input_par_function()


class ParametricSensitivityTaylorExpansion:
    """Reference: sensitivity_lib.py:1021-1363 -- the Taylor series of the optimum in the
    hyper-parameter.  The reference expands an arbitrary estimating equation to arbitrary order
    with nested ``autograd.make_jvp`` calls; this build covers what has a fixed kernel shape
    (SURVEY.md section 8f, rank 4): **data weights of a registered ``WeightedGLMObjective``,
    orders 1 to 8** (any order for the gaussian family, whose higher link derivatives vanish --
    the reference's own order-4 regression test runs).  Orders 1 to 3 are one fused streaming pass over X
    plus one Cholesky solve each:

        d1 = -H^{-1} X^T (r .* dw)
        d2 = -H^{-1} X^T scale (w ell''' t^2 + 2 dw ell'' t),   t = X d1

        d3 = -H^{-1} X^T scale (w (l4 t^3 + 3 l3 t t2) + 3 dw (l3 t^2 + l2 t2)),   t2 = X d2, lk = k-th link derivative

    and every higher order k is Faa di Bruno per row (``vt_glm_taylor_coeff``: partial Bell polynomials of
    ``zeta_j = X d_j``), two passes over X (``zeta_{k-1}`` and ``X^T c``) plus the solve:

        d_k = -H^{-1} X^T scale (w sum_{m=2..k} l_{m+1} B_{k,m}(zeta) + k dw sum_{m=1..k-1} l_{m+1} B_{k-1,m}(zeta))

    Same constructor / classmethod arguments and methods as the reference.  ``max_input_order`` truncates
    like the reference does (:1242-1249: a term whose number of partial derivatives w.r.t. the input exceeds it
    is dropped): with 1 only the ``dw l2`` terms survive, with 2 the ``l4`` term goes.  The objective
    is linear in the weights, so every term has hyper order <= 1 and ``max_hyper_order`` >= 1 changes nothing;
    ``forward_mode`` and ``force`` are accepted and moot."""

    @classmethod
    def optimization_objective(cls, objective_function, input_val0, hyper_val0, order, hess0=None,
                               forward_mode=True, max_input_order=None, max_hyper_order=None,
                               force=False):
        # Reference :1031-1080.
        from .objectives import WeightedGLMObjective
        if not isinstance(objective_function, WeightedGLMObjective):
            raise NotImplementedError(
                'ParametricSensitivityTaylorExpansion needs a WeightedGLMObjective '
                '(autograd is not part of this build).')
        if hess0 is None:
            hess0 = objective_function.hessian(dev.to_device(input_val0), dev.to_device(hyper_val0))
        hess_solver = solver_lib.get_cholesky_solver(hess0)
        return cls(estimating_equation=_GradOf(objective_function), input_val0=input_val0,
                   hyper_val0=hyper_val0, order=order, hess_solver=hess_solver,
                   forward_mode=forward_mode, max_input_order=max_input_order,
                   max_hyper_order=max_hyper_order, force=force)

    def __init__(self, estimating_equation, input_val0, hyper_val0, order, hess_solver,
                 forward_mode=True, max_input_order=None, max_hyper_order=None, force=False):
        from . import rows
        from .objectives import WeightedGLMObjective
        obj = getattr(estimating_equation, '_obj', None)
        if not isinstance(obj, WeightedGLMObjective):
            raise NotImplementedError(
                'estimating_equation must be the gradient of a WeightedGLMObjective '
                '(autograd is not part of this build).')
        if max_input_order is not None and max_input_order < 1:
            raise ValueError('max_input_order must be >= 1.')
        if max_hyper_order is not None and max_hyper_order < 1:
            raise ValueError('max_hyper_order must be >= 1.')
        if order < 1:
            raise ValueError('order must be at least one.')
        if order > rows.TAYLOR_MAX_ORDER and obj.family != 'gaussian':
            raise NotImplementedError(
                'Taylor orders above %d are built for the gaussian family only (its higher link '
                'derivatives vanish).' % rows.TAYLOR_MAX_ORDER)
        self._obj = obj
        self._input_val0 = deepcopy(input_val0)
        self._hyper_val0 = deepcopy(hyper_val0)
        self._objective_function_eta_grad = estimating_equation
        self._order = order
        self._max_input_order = max_input_order
        self._max_hyper_order = max_hyper_order
        self._forward_mode = forward_mode
        self.hess_solver = hess_solver
        self._theta = dev.to_device(input_val0, obj.device).contiguous()
        self._w = dev.to_device(hyper_val0, obj.device).contiguous()
        if self._theta.shape != (obj.D,) or self._w.shape != (obj.N,):
            raise ValueError('input_val0 / hyper_val0 do not match the objective.')
        # per-row linear predictor and first link derivative at the base point (one pass, kept)
        self._z, self._r, _ = rows.glm_link(obj, self._theta, self._w, want_z=True)

    def get_max_order(self):
        return self._order

    def _solve(self, vec):
        chol = getattr(self.hess_solver, "cholesky", None)
        if chol is not None:
            return chol.solve(vec)
        return dev.to_device(self.hess_solver(dev.like_input(vec, self._input_val0)), vec.device)

    def _evaluate_dkinput_dhyperk(self, dhyper, input_derivs, k):
        # Reference :1209-1260 (same checks, same ``-1 * hess_solver(vec)``).
        from . import rows
        if k <= 0:
            raise ValueError('k must be at least one.')
        if k > self._order:
            raise ValueError('k must be no greater than the declared order={}'.format(self._order))
        if len(input_derivs) < k - 1:
            raise ValueError('Not enough eta_derivs provided.')
        obj = self._obj
        # terms with more than ``max_input_order`` partial derivatives w.r.t. the input are dropped
        # (reference :1247-1249); the gaussian family has none beyond order 1
        mio = self._max_input_order if self._max_input_order is not None else k
        if k == 1:
            vec = rows.gemv_t2(obj.X, self._r, dhyper)
        elif k == 2:
            # G_tt[d1, d1] (input order 2: the w l3 t^2 part) + 2 G_tw[d1, dw] (input order 1)
            w_eff = self._w if mio >= 2 else torch.zeros_like(self._w)
            vec = rows.glm_taylor2_rhs(obj, self._z, w_eff, dhyper, input_derivs[0])
        elif k == 3 and obj.family != 'gaussian':
            # A = w l4 t^3 (input order 3), B = 3 w l3 t t2 and C = 3 dw l3 t^2 (order 2), E = 3 dw l2 t2
            # (order 1); the kernel returns A + B + C + E for the inputs it is given
            d1, d2 = input_derivs[0], input_derivs[1]
            zero_n = torch.zeros_like(self._w)
            if mio >= 3:
                vec = rows.glm_taylor3_rhs(obj, self._z, self._w, dhyper, d1, d2)
            elif mio == 2:      # everything but A = kernel(w, dw = 0, t2 = 0)
                vec = rows.glm_taylor3_rhs(obj, self._z, self._w, dhyper, d1, d2) - \
                    rows.glm_taylor3_rhs(obj, self._z, self._w, zero_n, d1, zeta2=zero_n)
            else:               # E alone = kernel(0, dw, t2) - kernel(0, dw, 0)
                vec = rows.glm_taylor3_rhs(obj, self._z, zero_n, dhyper, d1, d2) - \
                    rows.glm_taylor3_rhs(obj, self._z, zero_n, dhyper, d1, zeta2=zero_n)
        elif obj.family != 'gaussian':
            # any higher order: Faa di Bruno per row (partial Bell polynomials of zeta_j = X d_j), two passes over X;
            # the zeta_j of the derivatives already found are kept for the next order
            # (kept by identity: the cache holds the derivative tensors themselves, so no address is ever reused)
            cache = getattr(self, '_zeta_cache', {})
            zetas = {j: c[2] for j, c in cache.items()
                     if j < k and c[0] is input_derivs[j - 1] and c[1] == input_derivs[j - 1]._version}
            vec = rows.glm_taylor_rhs(obj, self._z, self._w, dhyper, input_derivs, k, max_input_order=mio, zetas=zetas)
            self._zeta_cache = {j: (input_derivs[j - 1], input_derivs[j - 1]._version, zj) for j, zj in zetas.items()}
        else:
            # gaussian family: G is bilinear in (theta, w), so the order-k equation is
            # H d_k + k (d^2 G / d theta d w)[d_{k-1}, dw] = 0: one fused pass X^T (scale dw .* (X d_{k-1}))
            dvec = dhyper if obj.scale == 1.0 else dhyper * obj.scale
            vec = float(k) * rows.glm_hvp(obj.X, dvec, input_derivs[k - 2])
        obj._all_reduce(vec)            # row-sharded objective: partial sums over the ranks' rows
        return -1 * self._solve(vec)

    def _get_default_max_order(self, max_order):
        # Reference :1262-1271.
        if max_order is None:
            return self._order
        if max_order <= 0:
            raise ValueError('max_order must be greater than zero.')
        if max_order > self._order:
            raise ValueError('max_order must be no greater than the order={}'.format(self._order))
        return max_order

    def _dev_derivs(self, dhyper, max_order):
        dh = dev.to_device(dhyper, self._obj.device).contiguous()
        if dh.shape != self._w.shape:
            raise ValueError('dhyper has the wrong shape.')
        derivs = []
        for k in range(1, max_order + 1):
            derivs.append(self._evaluate_dkinput_dhyperk(dh, derivs, k))
        return derivs

    def evaluate_input_derivs(self, dhyper, max_order=None):
        """List of the directional derivatives ``d^k input / d hyper^k [dhyper, ...]`` (:1273-1285)."""
        max_order = self._get_default_max_order(max_order)
        return [dev.like_input(d, self._input_val0) for d in self._dev_derivs(dhyper, max_order)]

    def evaluate_taylor_series_terms(self, new_hyper_val, add_offset=True, max_order=None):
        # Reference :1288-1301.
        max_order = self._get_default_max_order(max_order)
        zero = torch.zeros_like(self._theta)
        terms = [self._theta.clone() if add_offset else zero]
        dhyper = dev.to_device(new_hyper_val, self._obj.device) - self._w
        fact = 1.0
        for k, d in enumerate(self._dev_derivs(dhyper, max_order), start=1):
            fact *= k
            terms.append(d / fact)
        return [dev.like_input(t, self._input_val0) for t in terms]

    def print_terms(self, k=None):
        """Print the derivative terms of the series (reference :1346-1363), in the reference's format
        ``Order: o\tprefactor * eta[orders] * eps[order]``, for the terms this build evaluates."""
        if k is not None and k > self._order:
            raise ValueError('k must be no greater than order={}'.format(self._order))
        nonlin = self._obj.family != 'gaussian'
        # (prefactor, eta_orders, eps_order) of the terms of d^k G / d eps^k = 0 that do not vanish for an
        # objective linear in the weights; eta_orders[j] = how many factors d^{j+1} eta the term carries
        table = {1: [(1, [1], 0), (1, [0], 1)],
                 2: [(1, [0, 1], 0), (2, [1, 0], 1)] + ([(1, [2, 0], 0)] if nonlin else []),
                 3: [(1, [0, 0, 1], 0), (3, [0, 1, 0], 1)] +
                    ([(1, [3, 0, 0], 0), (3, [1, 1, 0], 0), (3, [2, 0, 0], 1)] if nonlin else [])}
        for order in range(self._order):
            if k is None or order == (k - 1):
                print('\nTerms for order {}:'.format(order + 1))
                terms = table.get(order + 1)
                if terms is None and nonlin:
                    terms = _taylor_terms(order + 1)
                if terms is None:       # gaussian family, order > 3: H d_k + k G_tw[d_{k-1}, dw]
                    eta = [0] * (order + 1)
                    eta[order] = 1
                    eta2 = [0] * (order + 1)
                    eta2[order - 1] = 1
                    terms = [(1, eta, 0), (order + 1, eta2, 1)]
                for pre, eta_orders, eps_order in terms:
                    if self._max_input_order is not None and sum(eta_orders) > self._max_input_order \
                            and eta_orders[-1] == 0:
                        continue
                    print('Order: {}\t{} * eta{} * eps[{}]'.format(order + 1, pre, eta_orders, eps_order))

    def evaluate_taylor_series(self, new_hyper_val, add_offset=True, max_order=None, sum_terms=True):
        # Reference :1303-1342 (which sums the terms whatever ``sum_terms`` says).
        terms = self.evaluate_taylor_series_terms(new_hyper_val, add_offset=add_offset, max_order=max_order)
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        return total


def _partitions(n, largest=None):
    """Integer partitions of n as lists of part multiplicities a[j - 1] = how many parts equal j."""
    largest = n if largest is None else largest
    if n == 0:
        yield []
        return
    for part in range(min(n, largest), 0, -1):
        for rest in _partitions(n - part, part):
            yield [part] + rest


def _taylor_terms(k):
    """(prefactor, eta_orders, eps_order) of every term of d^k G / d eps^k for an estimating equation linear in the
    hyper-parameter (Faa di Bruno): a partition of k with a_j parts equal to j carries k! / prod_j (j!^a_j a_j!) and
    the factors d^j eta, a_j times each; the terms with one hyper derivative are k times the partitions of k - 1."""
    from math import factorial
    out = []
    for eps, n, lead in ((0, k, 1), (1, k - 1, k)):
        for parts in _partitions(n):
            a = [0] * k
            for p_ in parts:
                a[p_ - 1] += 1
            coef = factorial(n)
            for j, aj in enumerate(a, start=1):
                coef //= factorial(j) ** aj * factorial(aj)
            out.append((lead * coef, a, eps))
    return out


class _GradOf:
    """``autograd.grad(objective, argnum=0)`` for a registered objective, carrying the
    Jacobian w.r.t. the hyper-parameter that the reference builds with
    ``autograd.jacobian(..., argnum=1)`` (sensitivity_lib.py:180-181)."""

    def __init__(self, objective):
        self._obj = objective

    def __call__(self, opt_par, hyper_par):
        obj = self._obj
        g = obj.grad(dev.to_device(opt_par), dev.to_device(hyper_par))
        return dev.like_input(g, opt_par)

    def jacobian_hyper(self, opt_par, hyper_par):
        return self._obj.cross_hessian(dev.to_device(opt_par), dev.to_device(hyper_par))
