"""The step BEFORE the hot path: find theta-hat on the device.

The reference's tests and notebook obtain the optimum with ``scipy.optimize.minimize``
(vittles/tests/test_sensitivity_lib.py:474-478; notebook cell 14) on the host and only
then build the linear approximation; at N = 10^6..10^7 rows that host optimiser is the
bottleneck.  Here the same kernels the sensitivity uses (fused link + gradient + Hessian,
Cholesky, solve, objective value) run a damped Newton iteration with every array resident
in HBM; when the objective is row-sharded the partial Hessians / gradients / values are
all-reduced exactly as on the sensitivity path, so every rank takes identical steps.
"""
import torch

from . import device as dev, rows, sharding
from .linalg import DeviceCholesky
from .objectives import WeightedGLMObjective


def newton_glm(objective, hyper_par, opt_par_init=None, max_iter=50, grad_tol=1e-9, step_tol=1e-13,
               armijo=1e-4, max_halvings=30):
    """Minimise ``objective(theta, hyper_par)`` over theta for a ``WeightedGLMObjective``.

    Newton steps ``theta <- theta - t H^{-1} g`` with backtracking (t = 1, 1/2, ...) on the
    objective value until the Armijo condition holds.  Stops when ``||g||_2 <= grad_tol``
    (the quantity ``validate_optimum`` tests, vittles/sensitivity_lib.py:203-215) or when
    the step is below ``step_tol * max(1, ||theta||)``.

    Returns ``(theta_hat, info)``: theta_hat in the container type of ``hyper_par`` (numpy
    or CUDA tensor), info = dict(iterations, grad_norm, value, converged, halvings).
    """
    if not isinstance(objective, WeightedGLMObjective):
        raise TypeError('newton_glm needs a WeightedGLMObjective.')
    dev.require_gpu()
    w = dev.to_device(hyper_par, objective.device).contiguous()
    D = objective.D
    if opt_par_init is None:
        theta = torch.zeros(D, dtype=torch.float64, device=objective.device)
    else:
        theta = dev.to_device(opt_par_init, objective.device).clone().contiguous()
        if theta.shape != (D,):
            raise ValueError('opt_par_init must have shape (%d,)' % D)
    group = objective.process_group

    def agree(*vals):           # every rank branches on rank 0's numbers (see sharding.agree_scalars)
        return sharding.agree_scalars(vals, group, objective.device)

    (value,) = agree(objective.value(theta, w))
    halvings_total = 0
    converged = False
    gnorm = float('inf')
    it = 0
    for it in range(1, max_iter + 1):
        out = objective.fused(theta, w)                     # K0 + K1 + K3 (+ all-reduce)
        g, H = out["g"], out["H"]
        if objective.dtype == torch.float32:
            g = objective.grad(theta, w)                    # fp64-summed gradient of the float data
        (gnorm,) = agree(rows.norm(g))
        if gnorm <= grad_tol:
            converged = True
            it -= 1
            break
        step = DeviceCholesky.factor(H, pack=True).solve(g)  # K4 + K5; LinAlgError if H is not PD
        slope, step_norm, theta_norm = agree(float(rows.dot(g, step).item()),            # slope > 0 for PD H
                                             rows.norm(step), rows.norm(theta))
        # objective values carry rounding noise of a few ulps of |f|: once the predicted decrease
        # is below it the value test says nothing, and the full Newton step is taken
        noise = 64.0 * 2.2e-16 * max(1.0, abs(value))
        t = 1.0
        for _ in range(max_halvings + 1):
            cand = theta - t * step
            (cand_value,) = agree(objective.value(cand, w))
            if cand_value == cand_value and cand_value <= value - armijo * t * slope + noise:
                break
            t *= 0.5
            halvings_total += 1
        else:
            break                                            # no decrease along the Newton direction
        snorm = t * step_norm
        theta, value = cand, cand_value
        if snorm <= step_tol * max(1.0, theta_norm):
            (gnorm,) = agree(rows.norm(objective.grad(theta, w)))
            converged = gnorm <= max(grad_tol, 1e-6)
            break
    info = dict(iterations=it, grad_norm=gnorm, value=value, converged=converged, halvings=halvings_total)
    return dev.like_input(theta, hyper_par), info
