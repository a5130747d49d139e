"""KPLS input rotations on the host (crates/gp/src/algorithm.rs:843-855): the reference takes them from linfa-pls 0.8.0
(`PlsRegression::params(k).fit(..).rotations().0`, a port of scikit-learn's NIPALS PLS regression, un-vendored).  For the
single-output case the NIPALS inner loop has a closed form per component; restated here in numpy so that
`.kpls_dim(k)` works without a user-supplied `w_star`.  The kernels only see |w| or w^2, so the sign convention of
the columns is irrelevant.  O(n d k) host work, once per fit; the GP itself runs on the GPU.
"""
from __future__ import annotations

import numpy as np


def pls_rotations(x, y, n_components):
    """x (n, d), y (n,) -> x rotations (d, k) of a PLS1 regression on centred, unit-variance (ddof = 1) data.
    A constant residual (e.g. a constant y) gives zeros, as the reference does on PowerMethodConstantResidualError."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n, d = x.shape
    k = int(n_components)
    xs = x.std(axis=0, ddof=1)
    xs[xs == 0.0] = 1.0
    ys = y.std(ddof=1)
    if ys <= 1e-12 * max(1.0, abs(float(y.mean()))):  # a constant response up to rounding
        return np.zeros((d, k))
    xk = (x - x.mean(axis=0)) / xs
    yk = (y - y.mean()) / (ys if ys != 0.0 else 1.0)
    w = np.zeros((d, k))
    p = np.zeros((d, k))
    eps = np.finfo(np.float64).eps
    for a in range(k):
        if np.all(np.abs(yk) < 10 * eps):  # constant residual
            return np.zeros((d, k))
        wa = xk.T @ yk
        nrm = np.linalg.norm(wa)
        if nrm < eps:
            return np.zeros((d, k))
        wa /= nrm
        t = xk @ wa
        tt = t @ t
        pa = xk.T @ t / tt
        qa = yk @ t / tt
        xk = xk - np.outer(t, pa)
        yk = yk - qa * t
        w[:, a], p[:, a] = wa, pa
    return w @ np.linalg.pinv(p.T @ w)
