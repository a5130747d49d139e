"""CPU oracle for the sparse Gaussian process (FITC / VFE) -- TEST INFRASTRUCTURE ONLY.

Restates crates/gp/src/sparse_algorithm.rs with numpy/scipy:
    compute_k           :218-234, 676-693   K(a, b) = sigma2 * corr(pairwise_differences(a, b), theta, w)
    fitc                :695-766            reduced likelihood + Woodbury data, FITC
    vfe                 :769-831            reduced likelihood + Woodbury data, VFE
    predict             :237-241            Kx . vec
    predict_var         :245-257            sigma2 - sum((inv^T Kx) * Kx) clamped at 1e-15, + noise
Unlike the full GP, the sparse GP works on RAW x / y (no normalisation) and has no trend (:141-144, 425-440).

PARITY UNPINNED against reference values: the reference's only printed SGP numbers (doc/SparseGpx_Tutorial.ipynb
cell 16) depend on inducing points drawn with Rust's Xoshiro256Plus shuffle, which cannot be reproduced here, and
its unit tests assert prediction error bounds on randomly drawn data.  The restatement is instead pinned
MATHEMATICALLY (tests/test_sgp_cpu.py): FITC's reduced likelihood equals the dense Gaussian log-density of
N(0, Qnn + diag(nu)) minus its constant, VFE's equals the dense variational bound, and the Woodbury predictions equal
the dense posterior formulas.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

from . import gp_oracle as O

FITC = "Fitc"
VFE = "Vfe"
DEFAULT_NOISE = 1e-2                         # ParamTuning::default init, sparse_parameters.rs:33-39
DEFAULT_NOISE_BOUNDS = (100.0 * np.finfo(float).eps, 1e10)
DEFAULT_THETA_BOUNDS = (1e-2, 1e2)           # sparse_parameters.rs:160-163


def compute_k(corr, a, b, w_star, theta, sigma2):
    dx = O.pairwise_differences(a, b)
    r = O.corr_value(corr, dx, theta, w_star)
    return r.reshape(a.shape[0], b.shape[0]) * sigma2


def _tri_inv(lower):
    return sla.solve_triangular(lower, np.eye(lower.shape[0]), lower=True, check_finite=False)


def fitc(corr, theta, sigma2, noise, w_star, xtrain, ytrain, z, nugget):
    """sparse_algorithm.rs:695-766 -> (likelihood, (vec, inv))."""
    nz = z.shape[0]
    knn = np.full(xtrain.shape[0], sigma2)
    kmm = compute_k(corr, z, z, w_star, theta, sigma2) + np.eye(nz) * nugget
    kmn = compute_k(corr, z, xtrain, w_star, theta, sigma2)
    u = np.linalg.cholesky(kmm)
    ui = _tri_inv(u)
    v = ui.dot(kmn)
    nu = knn - (v * v).sum(axis=0) + noise
    beta = 1.0 / nu
    a = np.eye(nz) + (v * beta[None, :]).dot(v.T)
    l_ = np.linalg.cholesky(a)
    li = _tri_inv(l_)
    a_vec = ytrain * beta[:, None]
    b = li.dot(v).dot(a_vec)
    term1 = np.log(nu).sum()
    term2 = 2.0 * np.log(np.diag(l_)).sum()
    term3 = float(a_vec.T.dot(ytrain)[0, 0])
    term4 = -float((b * b).sum())
    likelihood = -0.5 * (term1 + term2 + term3 + term4)
    li_ui = li.dot(ui)
    return likelihood, (li_ui.T.dot(b), ui.T.dot(ui) - li_ui.T.dot(li_ui))


def vfe(corr, theta, sigma2, noise, w_star, xtrain, ytrain, z, nugget):
    """sparse_algorithm.rs:769-831."""
    nz = z.shape[0]
    n = ytrain.shape[0]
    kmm = compute_k(corr, z, z, w_star, theta, sigma2) + np.eye(nz) * nugget
    kmn = compute_k(corr, z, xtrain, w_star, theta, sigma2)
    u = np.linalg.cholesky(kmm)
    ui = _tri_inv(u)
    v = ui.dot(kmn)
    beta = 1.0 / max(noise, nugget)
    a = v.dot(v.T) * beta
    l_ = np.linalg.cholesky(np.eye(nz) + a)
    li = _tri_inv(l_)
    b = li.dot(v).dot(ytrain) * beta
    term1 = -n * np.log(beta)
    term2 = 2.0 * np.log(np.diag(l_)).sum()
    term3 = beta * float((ytrain * ytrain).sum())
    term4 = -float(b.T.dot(b)[0, 0])
    term5 = n * beta * sigma2
    term6 = -np.trace(a)
    likelihood = -0.5 * (term1 + term2 + term3 + term4 + term5 + term6)
    li_ui = li.dot(ui)
    bi = np.eye(nz) + li.T.dot(li)
    return likelihood, (li_ui.T.dot(b), ui.T.dot(bi).dot(ui))


class SparseGpOracle:
    """Fitted state of sparse_algorithm.rs:145-168 at GIVEN (theta, sigma2, noise, z)."""

    def __init__(self, x, y, z, theta, sigma2, noise, corr=O.SQEXP, method=FITC, nugget=O.DEFAULT_NUGGET, w_star=None):
        self.x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        if self.x.shape[0] == 1 and np.asarray(x).ndim == 1:
            self.x = self.x.T
        self.y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
        self.z = np.atleast_2d(np.asarray(z, dtype=np.float64))
        self.corr, self.method, self.nugget = corr, method, nugget
        self.w_star = np.eye(self.x.shape[1]) if w_star is None else np.asarray(w_star, dtype=np.float64)
        self.theta = O.expand_theta(theta, self.w_star.shape[1])
        self.sigma2, self.noise = float(sigma2), float(noise)
        fn = fitc if method == FITC else vfe
        self.likelihood, (self.w_vec, self.w_inv) = fn(corr, self.theta, self.sigma2, self.noise, self.w_star, self.x,
                                                       self.y, self.z, nugget)

    def predict(self, xq):
        kx = compute_k(self.corr, np.atleast_2d(xq), self.z, self.w_star, self.theta, self.sigma2)
        return kx.dot(self.w_vec)[:, 0]

    def predict_var(self, xq):
        xq = np.atleast_2d(xq)
        kx = compute_k(self.corr, self.z, xq, self.w_star, self.theta, self.sigma2)
        var = self.sigma2 - (self.w_inv.T.dot(kx) * kx).sum(axis=0)
        return np.where(var < 1e-15, 1e-15, var) + self.noise
