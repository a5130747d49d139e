"""CPU oracle for the mixture-of-experts recombination around the GP experts (TEST INFRASTRUCTURE ONLY).

Restates, with numpy/scipy, the predict-side of egobox-moe that calls the accelerated GP path
(SURVEY.md 8f rank 1): the Gaussian-mixture responsibilities and the smooth / hard recombination of the
experts' mean and variance.  Pinned by the reference's own known answers for `pdfs`
(crates/moe/src/gaussian_mixture.rs:378-398) in tests/test_moe_cpu.py.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla


class GaussianMixtureOracle:
    """crates/moe/src/gaussian_mixture.rs:60-300."""

    def __init__(self, weights, means, covariances, heaviside_factor=1.0):
        self.weights = np.asarray(weights, dtype=np.float64)
        self.means = np.asarray(means, dtype=np.float64)
        self.covariances = np.asarray(covariances, dtype=np.float64)
        k, nx = self.means.shape
        # compute_precisions_cholesky :182-205: (L^-1)^T per cluster
        self.precisions_chol = np.empty((k, nx, nx))
        for i in range(k):
            c = sla.cholesky(self.covariances[i], lower=True)
            self.precisions_chol[i] = sla.solve_triangular(c, np.eye(nx), lower=True).T
        self.heaviside_factor = float(heaviside_factor)
        self.log_det = self._log_det()

    def _log_det(self):
        # compute_log_det :221-227 + compute_log_det_cholesky :287-299
        precs = self.precisions_chol * self.heaviside_factor ** -0.5
        return np.array([np.log(np.diag(p)).sum() for p in precs])

    def log_gaussian_prob(self, x):
        # compute_log_gaussian_prob :256-283
        x = np.asarray(x, dtype=np.float64)
        n, nx = x.shape
        precs = self.precisions_chol * self.heaviside_factor ** -0.5
        lp = np.empty((n, self.means.shape[0]))
        for k in range(self.means.shape[0]):
            diff = (x - self.means[k]).dot(precs[k])
            lp[:, k] = (diff * diff).sum(axis=1)
        cst = nx * math.log(2.0 * math.pi)
        return -0.5 * (lp + cst) + self.log_det

    def log_prob_resp(self, x):
        # compute_log_prob_resp :231-252
        wlp = self.log_gaussian_prob(x) + np.log(self.weights)
        e = np.where(wlp <= -307.0, 0.0, np.exp(wlp))  # f64::MIN_10_EXP = -307
        s = e.sum(axis=1)
        norm = np.where(np.abs(s) < np.finfo(float).eps, 0.0, np.log(np.where(s > 0, s, 1.0)))
        return norm, wlp - norm[:, None]

    def predict_probas(self, x):
        # :114-121
        x = np.asarray(x, dtype=np.float64)
        if self.means.shape[0] == 1:
            return np.ones((x.shape[0], 1))
        return np.exp(self.log_prob_resp(x)[1])

    def predict(self, x):
        # predict_inplace :305-316
        return np.argmax(np.exp(self.log_prob_resp(np.asarray(x, dtype=np.float64))[1]), axis=1)

    def pdfs(self, x):
        # :172-176
        return np.exp(self.log_gaussian_prob(np.asarray(x, dtype=np.float64).reshape(1, -1))[0])

    def precisions(self):
        # compute_precisions :207-215
        return np.array([pc.dot(pc.T) for pc in self.precisions_chol])

    def predict_single_probas_derivatives(self, x):
        # :127-153 -> (k, nx)
        x = np.asarray(x, dtype=np.float64).reshape(-1)
        pdfs = self.pdfs(x)
        v = self.weights.dot(pdfs)
        precs = self.precisions() / self.heaviside_factor
        deriv = np.array([(x - mu).dot(prec) for mu, prec in zip(self.means, precs)])
        vprime = (deriv * (-self.weights * pdfs)[:, None]).sum(axis=0)
        u = (self.weights * pdfs)[:, None]
        uprime = -(deriv * u)
        return (uprime * v - u * vprime[None, :]) / (v * v)

    def predict_probas_derivatives(self, x):
        # :158-170 -> (m, k, nx)
        x = np.asarray(x, dtype=np.float64)
        return np.array([self.predict_single_probas_derivatives(xi) for xi in x])


# ---- recombination, crates/moe/src/algorithm.rs ------------------------------------------------
def predict_smooth(experts, gmx, x):
    """:411-423."""
    p = gmx.predict_probas(x)
    return sum(e.predict(x) * p[:, i] for i, e in enumerate(experts))


def predict_var_smooth(experts, gmx, x):
    """:670-685  sum_i var_i p_i^2."""
    p = gmx.predict_probas(x)
    return sum(e.predict_var(x) * p[:, i] * p[:, i] for i, e in enumerate(experts))


def predict_hard(experts, gmx, x):
    """:879-888: one expert call per row with a 1 x nx batch."""
    x = np.asarray(x, dtype=np.float64)
    c = gmx.predict(x)
    return np.array([experts[c[i]].predict(x[i:i + 1])[0] for i in range(x.shape[0])])


def predict_var_hard(experts, gmx, x):
    """:894-910."""
    x = np.asarray(x, dtype=np.float64)
    c = gmx.predict(x)
    return np.array([experts[c[i]].predict_var(x[i:i + 1])[0] for i in range(x.shape[0])])


def predict_gradients_smooth(experts, gmx, x):
    """:691-733  sum_i p_i grad y_i + sum_i p'_i y_i, point by point."""
    x = np.asarray(x, dtype=np.float64)
    p, pp = gmx.predict_probas(x), gmx.predict_probas_derivatives(x)
    out = np.zeros_like(x)
    for a in range(x.shape[0]):
        xa = x[a:a + 1]
        preds = np.array([e.predict(xa)[0] for e in experts])
        drvs = np.array([e.predict_gradients(xa)[0] for e in experts])
        out[a] = (drvs * p[a][:, None]).sum(axis=0) + (pp[a] * preds[:, None]).sum(axis=0)
    return out


def predict_var_gradients_smooth(experts, gmx, x):
    """:739-783  sum_i p_i^2 grad v_i + 2 sum_i p_i p'_i v_i."""
    x = np.asarray(x, dtype=np.float64)
    p, pp = gmx.predict_probas(x), gmx.predict_probas_derivatives(x)
    out = np.zeros_like(x)
    for a in range(x.shape[0]):
        xa = x[a:a + 1]
        preds = np.array([e.predict_var(xa)[0] for e in experts])
        drvs = np.array([e.predict_var_gradients(xa)[0] for e in experts])
        out[a] = (drvs * (p[a] ** 2)[:, None]).sum(axis=0) + (2.0 * p[a][:, None] * pp[a] * preds[:, None]).sum(axis=0)
    return out


def predict_gradients_hard(experts, gmx, x):
    """:942-960."""
    x = np.asarray(x, dtype=np.float64)
    c = gmx.predict(x)
    return np.array([experts[c[i]].predict_gradients(x[i:i + 1])[0] for i in range(x.shape[0])])


def predict_var_gradients_hard(experts, gmx, x):
    """:965-983."""
    x = np.asarray(x, dtype=np.float64)
    c = gmx.predict(x)
    return np.array([experts[c[i]].predict_var_gradients(x[i:i + 1])[0] for i in range(x.shape[0])])
