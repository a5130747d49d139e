"""The sparse-GP oracle (oracle/sgp_oracle.py) is PARITY-UNPINNED against reference numbers (its header says why);
these tests pin the restated algebra mathematically: the Woodbury forms the reference uses must equal the dense
definitions of FITC (Snelson & Ghahramani) and VFE (Titsias) on problems small enough to write them densely."""
import math

import numpy as np
import pytest

from oracle import gp_oracle as O
from oracle import sgp_oracle as S


def _problem(n=40, nz=7, d=2, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d)) * 2 - 1
    y = np.sin(3 * x[:, 0]) + 0.5 * x[:, 1] ** 2 + 0.05 * rng.standard_normal(n)
    z = x[rng.permutation(n)[:nz]].copy()
    return x, y, z


def _dense(corr, x, z, theta, sigma2, nugget):
    w = np.eye(x.shape[1])
    kmm = S.compute_k(corr, z, z, w, theta, sigma2) + nugget * np.eye(z.shape[0])
    kmn = S.compute_k(corr, z, x, w, theta, sigma2)
    qnn = kmn.T.dot(np.linalg.solve(kmm, kmn))
    return kmm, kmn, qnn


@pytest.mark.parametrize("corr", [O.SQEXP, O.ABSEXP, O.MATERN32, O.MATERN52])
def test_fitc_equals_dense_gaussian_density(corr):
    x, y, z = _problem()
    theta, sigma2, noise, nugget = np.array([1.3, 0.8]), 0.9, 0.02, 1e-10
    sgp = S.SparseGpOracle(x, y, z, theta, sigma2, noise, corr=corr, method=S.FITC, nugget=nugget)
    kmm, kmn, qnn = _dense(corr, x, z, theta, sigma2, nugget)
    lam = sigma2 - np.diag(qnn) + noise
    cov = qnn + np.diag(lam)
    sign, logdet = np.linalg.slogdet(cov)
    dense = -0.5 * (logdet + y.dot(np.linalg.solve(cov, y)))
    assert sign > 0 and sgp.likelihood == pytest.approx(dense, rel=1e-9)
    # predictions: mean = K*m Sigma^-1 Kmn Lam^-1 y ; var = K** - K*m (Kmm^-1 - Sigma^-1) Km* + noise
    xq = np.random.default_rng(1).random((9, 2)) * 2 - 1
    ksm = S.compute_k(corr, xq, z, np.eye(2), theta, sigma2)
    sig = kmm + (kmn / lam).dot(kmn.T)
    mean = ksm.dot(np.linalg.solve(sig, (kmn / lam).dot(y)))
    var = sigma2 - np.einsum("ij,ij->i", ksm, np.linalg.solve(kmm, ksm.T).T - np.linalg.solve(sig, ksm.T).T) + noise
    np.testing.assert_allclose(sgp.predict(xq), mean, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(sgp.predict_var(xq), var, rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("corr", [O.SQEXP, O.MATERN52])
def test_vfe_equals_dense_variational_bound(corr):
    x, y, z = _problem(seed=3)
    theta, sigma2, noise, nugget = np.array([1.1, 0.6]), 1.2, 0.03, 1e-10
    sgp = S.SparseGpOracle(x, y, z, theta, sigma2, noise, corr=corr, method=S.VFE, nugget=nugget)
    kmm, kmn, qnn = _dense(corr, x, z, theta, sigma2, nugget)
    n = x.shape[0]
    cov = qnn + noise * np.eye(n)
    sign, logdet = np.linalg.slogdet(cov)
    dense = -0.5 * (logdet + y.dot(np.linalg.solve(cov, y))) - 0.5 / noise * (n * sigma2 - np.trace(qnn))
    assert sign > 0 and sgp.likelihood == pytest.approx(dense, rel=1e-9)
    xq = np.random.default_rng(2).random((5, 2)) * 2 - 1
    ksm = S.compute_k(corr, xq, z, np.eye(2), theta, sigma2)
    sig = kmm + kmn.dot(kmn.T) / noise
    mean = ksm.dot(np.linalg.solve(sig, kmn.dot(y) / noise))
    np.testing.assert_allclose(sgp.predict(xq), mean, rtol=1e-7, atol=1e-9)
    # The reference's VFE Woodbury matrix is  Ui^T (I + Li^T Li) Ui = Kmm^-1 + Sigma^-1  (sparse_algorithm.rs:824-828,
    # a PLUS where Titsias' posterior has a minus); parity means following the reference, clamp included (:249-255).
    var = sigma2 - np.einsum("ij,ij->i", ksm, np.linalg.solve(kmm, ksm.T).T + np.linalg.solve(sig, ksm.T).T)
    var = np.where(var < 1e-15, 1e-15, var) + noise
    np.testing.assert_allclose(sgp.predict_var(xq), var, rtol=1e-7, atol=1e-9)


def test_inducing_points_equal_training_points_recovers_the_noisy_full_gp():
    """z = x: Qnn = Knn, FITC's nu = noise -> the exact GP with a noise term."""
    x, y, _ = _problem(n=25)
    theta, sigma2, noise = np.array([0.9, 1.4]), 1.0, 0.05
    sgp = S.SparseGpOracle(x, y, x.copy(), theta, sigma2, noise, method=S.FITC, nugget=1e-12)
    k = S.compute_k(O.SQEXP, x, x, np.eye(2), theta, sigma2)
    cov = k + noise * np.eye(25)
    dense = -0.5 * (np.linalg.slogdet(cov)[1] + y.dot(np.linalg.solve(cov, y)))
    assert sgp.likelihood == pytest.approx(dense, rel=1e-6)
    xq = x[:4] + 0.01
    ks = S.compute_k(O.SQEXP, xq, x, np.eye(2), theta, sigma2)
    np.testing.assert_allclose(sgp.predict(xq), ks.dot(np.linalg.solve(cov, y)), rtol=1e-5, atol=1e-7)
    assert math.isfinite(sgp.likelihood)
