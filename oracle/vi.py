"""ORACLE (test infrastructure, never a product path): closed-form derivatives of
the registered hierarchical mean-field Gaussian VI objective (BASELINE.json
configs[4]).  The reference has no such model: it would differentiate the same
callable with autograd (lr_cov_lib.py:77-80; sparse_hessian_lib.py:59-60); no
reference test pins its numbers ("parity unpinned" beyond closed form == torch
AD and the exactness property LR-cov == posterior covariance that
tests/test_lr_cov_lib.py:90-93 uses).

Model (G groups, ng rows each, k coefficients per group):
    y_gi ~ N(x_gi . theta_g, 1/tau_y),  theta_g ~ N(mu, I/tau_th),  mu ~ N(0, I/tau_mu)
    q(theta_g) = N(m_g, diag(v_g)),  q(mu) = N(m_mu, diag(v_mu))
opt_par layout: [m_0, v_0, m_1, v_1, ..., m_{G-1}, v_{G-1}, m_mu, v_mu], d = 2k(G+1).
objective(eta, w) = KL(q || p(.|y)) up to a constant, data rows weighted by w_gi.
"""
import numpy as np


class HierGaussMFVI:
    def __init__(self, X, y, G, ng, k, tau_y=1.0, tau_th=1.0, tau_mu=1.0):
        self.X = np.asarray(X, dtype=np.float64).reshape(G, ng, k)
        self.y = np.asarray(y, dtype=np.float64).reshape(G, ng)
        self.G, self.ng, self.k = G, ng, k
        self.tau_y, self.tau_th, self.tau_mu = tau_y, tau_th, tau_mu
        self.d = 2 * k * (G + 1)

    # --- layout helpers -------------------------------------------------
    def unpack(self, eta):
        G, k = self.G, self.k
        loc = eta[:2 * k * G].reshape(G, 2, k)
        return loc[:, 0, :], loc[:, 1, :], eta[2 * k * G:2 * k * G + k], eta[2 * k * G + k:]

    def pack(self, m, v, mmu, vmu):
        return np.concatenate([np.stack([m, v], axis=1).reshape(-1), mmu, vmu])

    def sparsity_array(self):
        G, k = self.G, self.k
        return np.arange(2 * k * G).reshape(G, 2 * k)

    def global_inds(self):
        return np.arange(2 * self.k * self.G, self.d)

    # --- derivatives ----------------------------------------------------
    def value(self, eta, w=None):
        m, v, mmu, vmu = self.unpack(eta)
        w = np.ones((self.G, self.ng)) if w is None else np.asarray(w).reshape(self.G, self.ng)
        resid = self.y - np.einsum('gik,gk->gi', self.X, m)
        lik = 0.5 * self.tau_y * np.sum(w * (resid ** 2 + np.einsum('gik,gk->gi', self.X ** 2, v)))
        hier = 0.5 * self.tau_th * (np.sum((m - mmu) ** 2) + np.sum(v) + self.G * np.sum(vmu))
        prior = 0.5 * self.tau_mu * (np.sum(mmu ** 2) + np.sum(vmu))
        ent = 0.5 * np.sum(np.log(v)) + 0.5 * np.sum(np.log(vmu))
        return lik + hier + prior - ent

    def grad(self, eta, w=None):
        m, v, mmu, vmu = self.unpack(eta)
        w = np.ones((self.G, self.ng)) if w is None else np.asarray(w).reshape(self.G, self.ng)
        resid = self.y - np.einsum('gik,gk->gi', self.X, m)
        gm = -self.tau_y * np.einsum('gik,gi->gk', self.X, w * resid) + self.tau_th * (m - mmu)
        gv = 0.5 * self.tau_y * np.einsum('gik,gi->gk', self.X ** 2, w) + 0.5 * self.tau_th - 0.5 / v
        gmmu = -self.tau_th * np.sum(m - mmu, axis=0) + self.tau_mu * mmu
        gvmu = 0.5 * (self.G * self.tau_th + self.tau_mu) - 0.5 / vmu
        return self.pack(gm, gv, gmmu, gvmu)

    def hvp(self, eta, vec, w=None):
        m, v, mmu, vmu = self.unpack(eta)
        um, uv, ummu, uvmu = self.unpack(vec)
        w = np.ones((self.G, self.ng)) if w is None else np.asarray(w).reshape(self.G, self.ng)
        t = np.einsum('gik,gk->gi', self.X, um)
        hm = self.tau_y * np.einsum('gik,gi->gk', self.X, w * t) + self.tau_th * (um - ummu)
        hv = 0.5 / v ** 2 * uv
        hmmu = -self.tau_th * np.sum(um, axis=0) + (self.G * self.tau_th + self.tau_mu) * ummu
        hvmu = 0.5 / vmu ** 2 * uvmu
        return self.pack(hm, hv, hmmu, hvmu)

    def hessian(self, eta, w=None):
        """Dense (d, d) Hessian (small G only)."""
        H = np.empty((self.d, self.d))
        e = np.zeros(self.d)
        for j in range(self.d):
            e[j] = 1.0
            H[:, j] = self.hvp(eta, e, w)
            e[j] = 0.0
        return H

    def optimum(self, w=None):
        """Exact minimiser: variances in closed form, means from the (Gaussian)
        posterior-mean linear system."""
        G, k = self.G, self.k
        w = np.ones((G, self.ng)) if w is None else np.asarray(w).reshape(G, self.ng)
        v = 1.0 / (self.tau_y * np.einsum('gik,gi->gk', self.X ** 2, w) + self.tau_th)
        vmu = np.full(k, 1.0 / (G * self.tau_th + self.tau_mu))
        A = self.tau_y * np.einsum('gik,gi,gil->gkl', self.X, w, self.X) + self.tau_th * np.eye(k)
        b = self.tau_y * np.einsum('gik,gi->gk', self.X, w * self.y)
        Ainv_b = np.linalg.solve(A, b[..., None])[..., 0]
        Ainv = np.linalg.inv(A)
        # Schur complement on m_mu: (G tau_th + tau_mu) I - tau_th^2 sum_g A_g^{-1}
        Smat = (G * self.tau_th + self.tau_mu) * np.eye(k) - self.tau_th ** 2 * Ainv.sum(axis=0)
        mmu = np.linalg.solve(Smat, self.tau_th * Ainv_b.sum(axis=0))
        m = Ainv_b + self.tau_th * np.einsum('gkl,l->gk', Ainv, mmu)
        return self.pack(m, v, mmu, vmu)


def synth(G, ng, k, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((G * ng, k))
    mu = rng.standard_normal(k)
    theta = mu + rng.standard_normal((G, k))
    y = np.einsum('gik,gk->gi', X.reshape(G, ng, k), theta) + rng.standard_normal((G, ng))
    return X, y.reshape(-1)
