"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only:
it reads /root/reference, which does not exist on the GPU box; the tests read only the
.npz files this script writes).

  solver_ref.npz   inputs of the reference's own solver test (tests/test_solver_lib.py:11-18,
                   legacy np.random.seed(101)) and the OUTPUTS OF THE REFERENCE ITSELF:
                   /root/reference/vittles/solver_lib.py loaded standalone (it needs only SciPy)
                   -> dense / prefactored / sparse closures on a vector and on a (d, 7) matrix.
                   (get_cg_solver cannot run: SciPy >= 1.14 rejects atol='legacy'.)
  glm_small.npz    ridge-logistic (N, D) in {(64, 5), (1000, 20)}: X, y, w, theta-hat and
                   H, g, jac0 from an independent fp64 AD engine (torch.func, standing in for
                   autograd which is absent) and S = -solver(jac0) from the REFERENCE's
                   get_cholesky_solver.
  notebook_kat.npz the data-weight example of the reference's notebook
                   (docs/source/example_notebooks/mle_weight_sensitivity_example.ipynb cells 5-24):
                   data regenerated from its seed, Hessian / cross-Hessian of its loss in
                   paragami's free parameterisation by torch.func AD, and the numbers the
                   notebook itself PRINTED (cells 16 and 24) as the expected outputs.
"""
import importlib.util
import os

import numpy as np
import scipy.linalg
import scipy.sparse
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_solver_lib():
    spec = importlib.util.spec_from_file_location("ref_solver_lib", "/root/reference/vittles/solver_lib.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def solver_golden(ref):
    np.random.seed(101)
    d = 10
    h_dense = np.random.random((d, d))
    h_dense = h_dense + h_dense.T + d * np.eye(d)
    v = np.random.random(d)
    V = np.random.random((d, 7))
    h_sparse = scipy.sparse.csc_matrix(h_dense)
    h_chol = scipy.linalg.cho_factor(h_dense)
    np.savez(os.path.join(HERE, "solver_ref.npz"),
             h=h_dense, v=v, V=V, chol_c=h_chol[0], chol_lower=np.array(h_chol[1]),
             dense_v=ref.get_dense_cholesky_solver(h_dense)(v),
             dense_V=ref.get_dense_cholesky_solver(h_dense)(V),
             auto_v=ref.get_cholesky_solver(h_dense)(v),
             prefactored_v=ref.get_dense_cholesky_solver(None, h_chol)(v),
             sparse_v=ref.get_cholesky_solver(h_sparse)(v),
             sparse_V=ref.get_sparse_cholesky_solver(h_sparse)(V),
             np_solve_v=np.linalg.solve(h_dense, v))


def glm_golden(ref):
    out = {}
    for (N, D) in [(64, 5), (1000, 20)]:
        rng = np.random.default_rng(N + D)
        X = rng.standard_normal((N, D))
        theta_star = rng.standard_normal(D) / np.sqrt(D)
        y = (rng.random(N) < 1.0 / (1.0 + np.exp(-X @ theta_star))).astype(np.float64)
        w = 0.5 + rng.random(N)
        Xt, yt = torch.tensor(X), torch.tensor(y)

        def f(t, ww):
            z = Xt @ t
            return (ww * (torch.nn.functional.softplus(z) - yt * z)).sum() + 0.5 * (t ** 2).sum()
        theta = torch.zeros(D, dtype=torch.float64)
        wt = torch.tensor(w)
        for _ in range(30):
            g = torch.func.grad(f, 0)(theta, wt)
            H = torch.func.hessian(f, 0)(theta, wt)
            theta = theta - torch.linalg.solve(H, g)
        g = torch.func.grad(f, 0)(theta, wt).numpy()
        H = torch.func.hessian(f, 0)(theta, wt).numpy()
        jac0 = torch.func.jacrev(torch.func.grad(f, 0), 1)(theta, wt).numpy()
        S = -1 * ref.get_cholesky_solver(H)(jac0)       # sensitivity_lib.py:226 with the reference's solver
        tag = "%dx%d_" % (N, D)
        out.update({tag + "X": X, tag + "y": y, tag + "w": w, tag + "theta": theta.numpy(),
                    tag + "H": H, tag + "g": g, tag + "jac0": jac0, tag + "S": S})
    np.savez(os.path.join(HERE, "glm_small.npz"), **out)


def notebook_golden(ref):
    np.random.seed(42)
    num_obs = 1000
    true_sigma = np.eye(3) * np.diag(np.array([1, 2, 3])) + np.random.random((3, 3)) * 0.1
    true_sigma = 0.5 * (true_sigma + true_sigma.T)
    true_mu = np.array([0, 1, 2])
    x = np.random.multivariate_normal(mean=true_mu, cov=true_sigma, size=(num_obs,))
    mu = x.mean(0)
    xc = x - mu
    sig = xc.T @ xc / num_obs
    xt = torch.tensor(x)

    def fold(free):
        # paragami.PSDSymmetricMatrixPattern free form: lower Cholesky factor, log diagonal,
        # lower triangle vectorised row by row; then mu.
        rows = []
        idx = 0
        for i in range(3):
            row = []
            for j in range(3):
                if j < i:
                    row.append(free[idx]); idx += 1
                elif j == i:
                    row.append(torch.exp(free[idx])); idx += 1
                else:
                    row.append(torch.zeros((), dtype=torch.float64))
            rows.append(torch.stack(row))
        L = torch.stack(rows)
        return L @ L.T, free[6:9]

    def loss(free, w):   # notebook cell 7 + example_utils.get_normal_log_prob
        S, m = fold(free)
        c = xt - m
        return torch.sum(w * 0.5 * (torch.einsum('ni,ij,nj->n', c, torch.linalg.inv(S), c) + torch.logdet(S)))

    Lc = np.linalg.cholesky(sig)
    free0 = []
    for i in range(3):
        for j in range(i + 1):
            free0.append(np.log(Lc[i, j]) if i == j else Lc[i, j])
    free0 = np.array(free0 + list(mu))
    f0, w0 = torch.tensor(free0), torch.ones(num_obs, dtype=torch.float64)
    H = torch.func.hessian(loss, 0)(f0, w0).numpy()
    J = torch.func.jacrev(torch.func.grad(loss, 0), 1)(f0, w0).numpy()
    np.savez(os.path.join(HERE, "notebook_kat.npz"), x=x, free0=free0, H=H, J=J,
             # printed by the reference's notebook, cell 16 ("Optimal") and cell 24 ("Approximate LOO", n=10)
             printed_opt_sigma=np.array([[1.06683522, 0.07910048, 0.04229475],
                                         [0.07910048, 1.89297797, -0.02650233],
                                         [0.04229475, -0.02650233, 2.92376984]]),
             printed_opt_mu=np.array([-0.04469438, 1.03094019, 1.8551187]),
             printed_loo10_sigma=np.array([[1.06789931, 0.07906974, 0.04205564],
                                           [0.07906974, 1.89118719, -0.0359618],
                                           [0.04205564, -0.0359618, 2.90264779]]),
             printed_loo10_mu=np.array([-0.04475159, 1.02902066, 1.85020216]))


if __name__ == "__main__":
    ref = load_reference_solver_lib()
    solver_golden(ref)
    glm_golden(ref)
    notebook_golden(ref)
    print("wrote", [f for f in os.listdir(HERE) if f.endswith(".npz")])
