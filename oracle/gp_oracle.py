"""CPU oracle for the egobox-gp kriging hot path (TEST INFRASTRUCTURE ONLY).

This file is a numpy/scipy *restatement* of the reference's algorithm for the
path named by BASELINE.json `north_star` (fit at fixed theta / reduced
likelihood / predict / predict_var).  It is the checker the HIP path is
compared with; it is NOT part of the product:

    only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it.  The shipped package (`egobox_amd/`) never does.

Parity status: **pinned**.  `tests/test_oracle_golden.py` checks this file
against every known-answer value the reference holds for the path
(`tests/golden/*.json`, extracted by `tests/golden/make_golden.py` from
`doc/Gpx_Tutorial.ipynb:165-167,421`, `crates/gp/src/utils.rs:151-242`,
`crates/gp/src/correlation_models.rs:598-641,719-726`,
`crates/gp/src/mean_models.rs:170-178`, `python/egobox/tests/test_gpmix.py:37-53`).

The reference is Rust and cannot be compiled or imported in this image (no
cargo/rustc); its linear algebra lives in un-vendored crates
(linfa-linalg 0.2.1 `cholesky`/`solve_triangular`/`qr`/`svd`, or LAPACK through
ndarray-linalg 0.17.0 with the `blas` feature -- Cargo.lock).  Those are
standard factorizations with unique results up to rounding (and the QR sign),
restated here with scipy/LAPACK; the reference's own CI runs the same tests on
both of its backends.

Every function cites the reference lines it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.linalg as sla

# correlation kinds (names follow crates/gp/src/correlation_models.rs)
SQEXP = "SquaredExponential"
ABSEXP = "AbsoluteExponential"
MATERN32 = "Matern32"
MATERN52 = "Matern52"
# regression kinds (crates/gp/src/mean_models.rs)
CONSTANT = "Constant"
LINEAR = "Linear"
QUADRATIC = "Quadratic"

#: crates/gp/src/parameters.rs:118  -- nugget = 100 * f64::EPSILON
DEFAULT_NUGGET = 100.0 * np.finfo(np.float64).eps
#: crates/gp/src/parameters.rs:49-51
DEFAULT_THETA_INIT = 1e-1
DEFAULT_THETA_BOUNDS = (1e-2, 1e1)


class LikelihoodComputationError(Exception):
    """crates/gp/src/errors.rs:12 GpError::LikelihoodComputationError."""


class NotPositiveDefinite(Exception):
    """linfa_linalg::LinalgError::NotPositiveDefinite (crates/gp/src/errors.rs:19)."""


# --------------------------------------------------------------------------
# crates/gp/src/utils.rs
# --------------------------------------------------------------------------
def normalize(x):
    """crates/gp/src/utils.rs:45-54: column mean, SAMPLE std (ddof=1), 0-std -> 1."""
    x = np.asarray(x, dtype=np.float64)
    mean = x.mean(axis=0)
    std = x.std(axis=0, ddof=1) if x.shape[0] > 1 else np.full(x.shape[1], np.nan)
    std = np.where(std == 0.0, 1.0, std)
    return (x - mean) / std, mean, std


def diff_matrix(x):
    """crates/gp/src/utils.rs:80-104 DiffMatrix::_cross_diff.

    Returns (d, d_indices): d[(k,i) pair] = |x_k - x_i| component-wise, pairs in
    k-major upper-triangle order (0,1),(0,2)...(n-2,n-1).
    """
    x = np.asarray(x, dtype=np.float64)
    n, nx = x.shape
    npairs = n * (n - 1) // 2
    d = np.zeros((npairs, nx))
    idx = np.zeros((npairs, 2), dtype=np.int64)
    pos = 0
    for k in range(n - 1):
        cnt = n - k - 1
        idx[pos:pos + cnt, 0] = k
        idx[pos:pos + cnt, 1] = np.arange(k + 1, n)
        d[pos:pos + cnt, :] = x[k, :] - x[k + 1:n, :]
        pos += cnt
    return np.abs(d), idx


def pairwise_differences(x, y):
    """crates/gp/src/utils.rs:110-131: signed x_i - y_j, row index i*ny + j."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    assert x.shape[1] == y.shape[1]
    return (x[:, None, :] - y[None, :, :]).reshape(x.shape[0] * y.shape[0], x.shape[1])


# --------------------------------------------------------------------------
# crates/gp/src/correlation_models.rs  (value only)
# --------------------------------------------------------------------------
def corr_value(kind, d, theta, weights):
    """CorrelationModel::value(d (N,nx), theta (h,), weights (nx,h)) -> (N,1).

    sq-exp  : correlation_models.rs:91-104
    abs-exp : correlation_models.rs:185-196
    Matern32: correlation_models.rs:277-286 + _compute_r_factors :326-353
    Matern52: correlation_models.rs:446-455 + compute_r_factors  :497-523
    """
    d = np.asarray(d, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    w = np.asarray(weights, dtype=np.float64)
    if kind == SQEXP:
        theta_w = ((theta * w) ** 2).sum(axis=1)  # (nx,)
        r = (d ** 2).dot(theta_w)
        return np.exp(-0.5 * r).reshape(-1, 1)
    if kind == ABSEXP:
        theta_w = np.abs(w).dot(np.broadcast_to(theta, (w.shape[1],)))
        r = np.abs(d).dot(theta_w)
        return np.exp(-r).reshape(-1, 1)
    if kind in (MATERN32, MATERN52):
        theta_w = theta * np.abs(w)  # (nx,h)
        abs_d = np.abs(d)
        a = np.ones(d.shape[0])
        if kind == MATERN32:
            s = math.sqrt(3.0)
            for j in range(d.shape[1]):
                for l in range(theta_w.shape[1]):
                    if theta_w[j, l] == 0.0:
                        continue  # factor is exactly 1.0 (w = identity: the reference multiplies by it d*(h-1) times)
                    a *= 1.0 + s * theta_w[j, l] * abs_d[:, j]
        else:
            s = math.sqrt(5.0)
            c53 = 5.0 / 3.0
            for j in range(d.shape[1]):
                for l in range(theta_w.shape[1]):
                    v = theta_w[j, l]
                    if v == 0.0:
                        continue  # factor is exactly 1.0
                    a *= 1.0 + s * v * abs_d[:, j] + c53 * (v * v * d[:, j] * d[:, j])
        b = np.exp(-s * abs_d.dot(theta_w).sum(axis=1))
        return (a * b).reshape(-1, 1)
    raise ValueError(f"unknown correlation kind {kind!r}")


# --------------------------------------------------------------------------
# crates/gp/src/mean_models.rs  (value only)
# --------------------------------------------------------------------------
def regression_value(kind, x):
    """RegressionModel::value: Constant :42-44, Linear :68-71, Quadratic :97-104."""
    x = np.asarray(x, dtype=np.float64)
    n, nx = x.shape
    ones = np.ones((n, 1))
    if kind == CONSTANT:
        return ones
    if kind == LINEAR:
        return np.concatenate([ones, x], axis=1)
    if kind == QUADRATIC:
        res = np.concatenate([ones, x], axis=1)
        for k in range(nx):
            res = np.concatenate([res, x[:, k:] * x[:, k:k + 1]], axis=1)
        return res
    raise ValueError(f"unknown regression kind {kind!r}")


def _signum(v):
    """Rust f64::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives (numpy sign(0) would be 0)."""
    return np.copysign(1.0, v)


def corr_jacobian(kind, x, xtrain, theta, weights):
    """CorrelationModel::jacobian(x (nx,), xtrain (n,nx), theta, weights) -> (n,nx): d r(x, X_i) / d x_k.

    sq-exp :106-123, abs-exp :198-214, Matern32 `_jac_helper` :355-413, Matern52 `_jac_helper` :524-586
    (crates/gp/src/correlation_models.rs).  `differences(x, xtrain)` = x - xtrain rows (utils.rs).
    """
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    xt = np.asarray(xtrain, dtype=np.float64)
    theta = np.asarray(theta, dtype=np.float64).reshape(-1)
    w = np.asarray(weights, dtype=np.float64)
    d = x[None, :] - xt
    r = corr_value(kind, d, theta, w)  # (n,1)
    if kind == SQEXP:
        dtheta_w = -((theta * w) ** 2).sum(axis=1)
        return d * dtheta_w * r
    if kind == ABSEXP:
        dtheta_w = _signum(d) * (-(theta * np.abs(w)).sum(axis=1))
        return dtheta_w * r
    if kind in (MATERN32, MATERN52):
        q = math.sqrt(3.0) if kind == MATERN32 else math.sqrt(5.0)
        theta_w = theta * np.abs(w)  # (nx,h)
        abs_d, sign_d = np.abs(d), _signum(d)
        n, nx = d.shape

        def factor(v):
            return 1.0 + q * v if kind == MATERN32 else 1.0 + q * v + (5.0 / 3.0) * v * v

        # (entries of theta_w that are exactly zero -- every off-diagonal one when the weights are the identity -- are skipped:
        #  factor(0) == 1.0 and their derivative term is 0.0 * finite, so the reference's full loops give the same bits; without
        #  the skip the four nested loops are nx^4 vector operations per point, 8e9 at nx = 300)
        a = np.ones(n)
        for j in range(nx):
            for l in range(theta_w.shape[1]):
                if theta_w[j, l] == 0.0:
                    continue
                a = a * factor(theta_w[j, l] * abs_d[:, j])
        b = np.exp(-q * abs_d.dot(theta_w).sum(axis=1))
        db = -q * np.abs(w).dot(theta)[None, :] * sign_d * (a * b)[:, None]
        da = np.zeros((n, nx))
        for j in range(nx):
            for k in range(theta_w.shape[1]):
                if theta_w[j, k] == 0.0:
                    continue
                if kind == MATERN32:
                    deriv = q * theta_w[j, k] * sign_d[:, j]
                else:
                    deriv = (q * theta_w[j, k] * sign_d[:, j]
                             + (10.0 / 3.0) * theta_w[j, k] ** 2 * sign_d[:, j] * abs_d[:, j])
                term = np.ones(n)
                for p in range(nx):
                    for l in range(theta_w.shape[1]):
                        if (l != k or p != j) and theta_w[p, l] != 0.0:
                            term = term * factor(theta_w[p, l] * abs_d[:, p])
                da[:, j] += deriv * term
        return db + da * b[:, None]
    raise ValueError(f"unknown correlation kind {kind!r}")


def regression_jacobian(kind, x):
    """RegressionModel::jacobian(x (nx,)) -> (p,nx): Constant :50-52, Linear :76-81, Quadratic :110-128
    (crates/gp/src/mean_models.rs); row order = the columns of `regression_value`."""
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    nx = x.size
    if kind == CONSTANT:
        return np.zeros((1, nx))
    if kind == LINEAR:
        return np.concatenate([np.zeros((1, nx)), np.eye(nx)], axis=0)
    if kind == QUADRATIC:
        rows = [np.zeros((1, nx)), np.eye(nx)]
        for i in range(nx):  # columns x_i * x_j, j >= i
            blk = np.zeros((nx - i, nx))
            for c, j in enumerate(range(i, nx)):
                blk[c, i] += x[j]
                blk[c, j] += x[i]
            rows.append(blk)
        return np.concatenate(rows, axis=0)
    raise ValueError(f"unknown regression kind {kind!r}")


# --------------------------------------------------------------------------
# crates/gp/src/algorithm.rs
# --------------------------------------------------------------------------
@dataclass
class GpInnerParams:
    """crates/gp/src/algorithm.rs:47-60."""
    sigma2: float
    beta: np.ndarray      # (p,1)
    gamma: np.ndarray     # (n,1)
    r_chol: np.ndarray    # (n,n) lower
    ft: np.ndarray        # (n,p)
    ft_qr_r: np.ndarray   # (p,p) upper, positive diagonal (Appendix A.7 of SURVEY)


def _qr_pos(ft):
    """Thin QR with the sign convention of the reference's serialized models
    (positive diagonal of R; doc/Gpx_Tutorial.ipynb:421 ft_qr_r)."""
    q, r = np.linalg.qr(ft, mode="reduced")
    s = np.sign(np.diag(r))
    s[s == 0] = 1.0
    return q * s, (r.T * s).T


def assemble_r(rxx, d_indices, n, nugget):
    """algorithm.rs:997-1001: R = I*(1+nugget); scatter rxx into both triangles."""
    r_mx = np.eye(n) * (1.0 + nugget)
    r_mx[d_indices[:, 0], d_indices[:, 1]] = rxx[:, 0]
    r_mx[d_indices[:, 1], d_indices[:, 0]] = rxx[:, 0]
    return r_mx


def reduced_likelihood_from_r(fx, r_mx, ynorm, y_std0, keep_chol=True):
    """algorithm.rs:1003-1055 given the assembled correlation matrix R.

    Returns (likelihood, GpInnerParams).  Raises NotPositiveDefinite /
    LikelihoodComputationError exactly where the reference returns Err.
    `r_mx` is overwritten by its Cholesky factor.
    """
    n = r_mx.shape[0]
    try:
        r_chol = sla.cholesky(r_mx, lower=True, overwrite_a=True, check_finite=False)  # :1004
    except np.linalg.LinAlgError as e:
        raise NotPositiveDefinite(str(e)) from None
    if not np.all(np.isfinite(np.diag(r_chol))):
        raise NotPositiveDefinite("non finite pivot")
    # make sure the strict upper triangle is zero (LAPACK leaves the input there)
    r_chol = np.tril(r_chol)
    ft = sla.solve_triangular(r_chol, fx, lower=True, check_finite=False)              # :1006
    q, rq = _qr_pos(ft)                                                                # :1007
    sv = np.linalg.svd(rq, compute_uv=False)                                           # :1010
    cond_ft = sv[-1] / sv[0]
    if cond_ft < 1e-10:                                                                # :1012-1027
        sv_f = np.linalg.svd(fx, compute_uv=False)
        if sv_f[0] / sv_f[-1] > 1e15:
            raise LikelihoodComputationError(
                "F is too ill conditioned. Poor combination of regression model and observations.")
        raise LikelihoodComputationError("ft is too ill conditioned, try another theta again")
    yt = sla.solve_triangular(r_chol, ynorm, lower=True, check_finite=False)           # :1028
    beta = sla.solve_triangular(rq, q.T.dot(yt), lower=False, check_finite=False)      # :1030
    rho = yt - ft.dot(beta)                                                            # :1031
    rho_sqr = (rho * rho).sum(axis=0)                                                  # :1032
    gamma = sla.solve_triangular(r_chol.T, rho, lower=False, check_finite=False)       # :1034
    logdet = np.log10(np.diag(r_chol)).sum() * 2.0 / n                                 # :1039
    sigma2 = rho_sqr / n                                                               # :1042
    lkh = -n * (math.log10(sigma2.sum()) + logdet)                                     # :1043
    inner = GpInnerParams(
        sigma2=float(sigma2[0] * y_std0 * y_std0),                                     # :1048
        beta=beta, gamma=gamma, r_chol=r_chol if keep_chol else None, ft=ft, ft_qr_r=rq)
    return float(lkh), inner


def reduced_likelihood(fx, rxx, d_indices, n, ynorm, y_std0, nugget):
    """algorithm.rs:988-1056 (signature-compatible restatement)."""
    return reduced_likelihood_from_r(fx, assemble_r(rxx, d_indices, n, nugget), ynorm, y_std0)


def corr_matrix_dense(kind, xnorm, theta, w_star, nugget, chunk=512):
    """R assembled directly from normalised X without the (n(n-1)/2, d) table.

    Element-for-element the same arithmetic as diff_matrix -> corr_value ->
    assemble_r (checked in tests/test_oracle_golden.py), but in row chunks so
    that n = 16384, d = 32 fits in host memory.
    """
    xnorm = np.asarray(xnorm, dtype=np.float64)
    n = xnorm.shape[0]
    r_mx = np.empty((n, n))
    for i0 in range(0, n, chunk):
        i1 = min(n, i0 + chunk)
        dx = np.abs(pairwise_differences(xnorm[i0:i1], xnorm))
        r_mx[i0:i1, :] = corr_value(kind, dx, theta, w_star).reshape(i1 - i0, n)
    r_mx[np.arange(n), np.arange(n)] = 1.0 + nugget
    return r_mx


@dataclass
class GaussianProcessOracle:
    """Fitted state, crates/gp/src/algorithm.rs:174-192."""
    theta: np.ndarray
    likelihood: float
    inner: GpInnerParams
    w_star: np.ndarray
    xt_norm: np.ndarray
    x_mean: np.ndarray
    x_std: np.ndarray
    yt_norm: np.ndarray
    y_mean: np.ndarray
    y_std: np.ndarray
    mean: str
    corr: str
    nugget: float = DEFAULT_NUGGET
    training_data: tuple = field(default=None, repr=False)

    # algorithm.rs:372-380
    def _compute_correlation(self, xnorm):
        dx = pairwise_differences(xnorm, self.xt_norm)
        r = corr_value(self.corr, dx, self.theta, self.w_star)
        return r.reshape(xnorm.shape[0], self.xt_norm.shape[0])

    # algorithm.rs:330-369
    def _compute_rt_u(self, xnorm, corr):
        inn = self.inner
        rt = sla.solve_triangular(inn.r_chol, corr.T, lower=True, check_finite=False)
        rhs = inn.ft.T.dot(rt) - regression_value(self.mean, xnorm).T
        u = sla.solve_triangular(inn.ft_qr_r.T, rhs, lower=True, check_finite=False)
        return rt, u

    def _chunks(self, x, chunk):
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        for a in range(0, x.shape[0], chunk):
            yield x[a:a + chunk]

    def predict(self, x, chunk=1024):
        """algorithm.rs:253-263."""
        out = []
        for xc in self._chunks(x, chunk):
            xnorm = (xc - self.x_mean) / self.x_std
            f = regression_value(self.mean, xnorm)
            corr = self._compute_correlation(xnorm)
            y_ = f.dot(self.inner.beta) + corr.dot(self.inner.gamma)
            out.append((y_ * self.y_std + self.y_mean)[:, 0])
        return np.concatenate(out)

    def predict_var(self, x, chunk=1024):
        """algorithm.rs:267-279."""
        out = []
        for xc in self._chunks(x, chunk):
            xnorm = (xc - self.x_mean) / self.x_std
            corr = self._compute_correlation(xnorm)
            rt, u = self._compute_rt_u(xnorm, corr)
            mse = 1.0 - (rt * rt).sum(axis=0) + (u * u).sum(axis=0)
            mse = self.inner.sigma2 * mse
            out.append(np.where(mse < 0.0, 0.0, mse))
        return np.concatenate(out)

    def predict_valvar(self, x, chunk=1024):
        """algorithm.rs:282-307."""
        return self.predict(x, chunk), self.predict_var(x, chunk)

    def predict_gradients(self, x):
        """algorithm.rs:510-549 (`predict_jacobian` per row): d mean / d x, (m, nx)."""
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        out = np.empty((x.shape[0], self.xt_norm.shape[1]))
        for a in range(x.shape[0]):
            xnorm = (x[a] - self.x_mean) / self.x_std
            df_dx = regression_jacobian(self.mean, xnorm).T.dot(self.inner.beta)           # (nx,1)
            dr = corr_jacobian(self.corr, xnorm, self.xt_norm, self.theta, self.w_star)    # (n,nx)
            out[a] = ((df_dx + dr.T.dot(self.inner.gamma))[:, 0]) * self.y_std[0] / self.x_std
        return out

    def predict_var_gradients(self, x):
        """algorithm.rs:555-617 (`predict_var_gradients_single` per row): d variance / d x, (m, nx)."""
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        inn = self.inner
        f_mean = regression_value(self.mean, self.xt_norm)
        rho2 = sla.solve_triangular(inn.r_chol, f_mean, lower=True, check_finite=False)
        inv_kf = sla.solve_triangular(inn.r_chol.T, rho2, lower=False, check_finite=False)
        b_mat = f_mean.T.dot(inv_kf)
        rho3 = np.linalg.cholesky(b_mat)
        out = np.empty((x.shape[0], self.xt_norm.shape[1]))
        for a in range(x.shape[0]):
            xnorm = ((x[a] - self.x_mean) / self.x_std).reshape(1, -1)
            r = self._compute_correlation(xnorm).T                                          # (n,1)
            dr = corr_jacobian(self.corr, xnorm[0], self.xt_norm, self.theta, self.w_star)  # (n,nx)
            rho1 = sla.solve_triangular(inn.r_chol, r, lower=True, check_finite=False)
            inv_kr = sla.solve_triangular(inn.r_chol.T, rho1, lower=False, check_finite=False)
            p2 = inv_kr.T.dot(dr)                                                           # (1,nx)
            f_x = regression_value(self.mean, xnorm).T                                      # (p,1)
            a_mat = f_x.T - r.T.dot(inv_kf)                                                 # (1,p)
            inv_bat = sla.solve_triangular(rho3, a_mat.T, lower=True, check_finite=False)
            d_mat = sla.solve_triangular(rho3.T, inv_bat, lower=False, check_finite=False)  # (p,1)
            d_a = regression_jacobian(self.mean, xnorm[0]).T - dr.T.dot(inv_kf)             # (nx,p)
            p4 = d_mat.T.dot(d_a.T)                                                         # (1,nx)
            out[a] = (2.0 * (p4 - p2) / self.x_std * inn.sigma2)[0]
        return out

    def predict_valvar_gradients(self, x):
        """algorithm.rs:711-727."""
        return self.predict_gradients(x), self.predict_var_gradients(x)


def expand_theta(theta, dim):
    """algorithm.rs:829-838: a length-1 init is broadcast to `dim`."""
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    if theta.size == 1:
        return np.full(dim, theta[0])
    if theta.size != dim:
        raise ValueError(
            f"Initial guess for theta should be either 1-dim or dim of xtrain (w_star.ncols()), got {theta.size}")
    return theta.copy()


def prepare_training(x, y, mean=CONSTANT):
    """The theta-independent part of GpValidParams::fit, algorithm.rs:795-866."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
    xn, xm, xs = normalize(x)
    yn, ym, ys = normalize(y)
    fx = regression_value(mean, xn)
    return x, y, xn, xm, xs, yn, ym, ys, fx


def likelihood_at(x, y, theta, mean=CONSTANT, corr=SQEXP, nugget=DEFAULT_NUGGET, w_star=None,
                  dense=None):
    """One evaluation of the objective of algorithm.rs:880-897 WITHOUT the sign flip:
    returns (likelihood, status) with status in
    {0 ok, 1 not positive definite, 2 ft ill conditioned, 3 F ill conditioned, 4 NaN theta}."""
    x, y, xn, xm, xs, yn, ym, ys, fx = prepare_training(x, y, mean)
    n, nx = xn.shape
    if w_star is None:
        w_star = np.eye(nx)
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    if np.any(np.isnan(theta)):
        return math.inf, 4
    try:
        if dense is None:
            dense = n > 1500
        if dense:
            r_mx = corr_matrix_dense(corr, xn, theta, w_star, nugget)
            lkh, _ = reduced_likelihood_from_r(fx, r_mx, yn, ys[0], keep_chol=False)
        else:
            d, idx = diff_matrix(xn)
            rxx = corr_value(corr, d, theta, w_star)
            lkh, _ = reduced_likelihood(fx, rxx, idx, n, yn, ys[0], nugget)
        return lkh, 0
    except NotPositiveDefinite:
        return -math.inf, 1
    except LikelihoodComputationError as e:
        return -math.inf, (3 if str(e).startswith("F is") else 2)


def fit_fixed(x, y, theta, mean=CONSTANT, corr=SQEXP, nugget=DEFAULT_NUGGET, w_star=None,
              dense=None):
    """GpValidParams::fit with ThetaTuning::Fixed(theta): algorithm.rs:791-872, 966-978.

    This is the reference's own parity entry point (its tightest test,
    algorithm.rs:1758-1765, and Python `Gpx.builder(n_start=-1, theta_init=..)`,
    python/src/gp_mix.rs:202-208).
    """
    x, y, xn, xm, xs, yn, ym, ys, fx = prepare_training(x, y, mean)
    n, nx = xn.shape
    if w_star is None:
        w_star = np.eye(nx)
    # ThetaTuning::Fixed passes theta straight to `value` (len-1 broadcasts there,
    # algorithm.rs:869-872, correlation_models.rs:97)
    theta = np.atleast_1d(np.asarray(theta, dtype=np.float64))
    if dense is None:
        dense = n > 1500
    if dense:
        r_mx = corr_matrix_dense(corr, xn, theta, w_star, nugget)
        lkh, inner = reduced_likelihood_from_r(fx, r_mx, yn, ys[0])
    else:
        d, idx = diff_matrix(xn)
        rxx = corr_value(corr, d, theta, w_star)
        lkh, inner = reduced_likelihood(fx, rxx, idx, n, yn, ys[0], nugget)
    return GaussianProcessOracle(
        theta=theta, likelihood=lkh, inner=inner, w_star=w_star,
        xt_norm=xn, x_mean=xm, x_std=xs, yt_norm=yn, y_mean=ym, y_std=ys,
        mean=mean, corr=corr, nugget=nugget, training_data=(x, y[:, 0]))


# --------------------------------------------------------------------------
# New capability (no reference counterpart): theta-gradient of the likelihood.
# SURVEY.md Appendix A.12; validated by central finite differences in tests.
# --------------------------------------------------------------------------
def likelihood_grad(x, y, theta, mean=CONSTANT, corr=SQEXP, nugget=DEFAULT_NUGGET):
    """dL/dtheta_k = (1/ln10) [ gamma^T (d_k R) gamma / sigma2 - tr(R^-1 d_k R) ]  (w = I)."""
    # (the reference's pairwise-difference layout needs n^2 d / 2 doubles: 34 GB at n = 16384, d = 32 -- the dense
    #  construction of R there, the same numbers to rounding)
    gp = fit_fixed(x, y, theta, mean, corr, nugget, dense=np.shape(x)[0] > 4096)
    xn = gp.xt_norm
    n, nx = xn.shape
    theta = expand_theta(theta, nx)
    r_mx = gp.inner.r_chol.dot(gp.inner.r_chol.T)
    # R^-1 = C^-T C^-1 through LAPACK's in-place triangular inverse (no n x n identity, no second copy), and the
    # per-dimension derivative matrices one block of rows at a time: at n = 16384 the whole job stays under ~12 GB
    cinv, info = sla.lapack.dtrtri(gp.inner.r_chol, lower=1, overwrite_c=0)
    if info != 0:
        raise np.linalg.LinAlgError(f"dtrtri info {info}")
    rinv = cinv.T.dot(cinv)
    del cinv
    gamma = gp.inner.gamma[:, 0]
    sigma2n = gp.inner.sigma2 / (gp.y_std[0] ** 2)
    grad = np.zeros(nx)
    rows = 1024
    for k in range(nx):
        quad, trace = 0.0, 0.0
        for i0 in range(0, n, rows):
            i1 = min(n, i0 + rows)
            a = np.abs(xn[i0:i1, None, k] - xn[None, :, k])
            rb = r_mx[i0:i1]
            if corr == SQEXP:
                dr = -theta[k] * a * a * rb
            elif corr == MATERN52:
                s5 = math.sqrt(5.0)
                t = theta[k] * a
                p = 1.0 + s5 * t + (5.0 / 3.0) * t * t
                dr = rb * ((s5 * a + (10.0 / 3.0) * theta[k] * a * a) / p - s5 * a)
            elif corr == MATERN32:
                s3 = math.sqrt(3.0)
                t = theta[k] * a
                dr = rb * (s3 * a / (1.0 + s3 * t) - s3 * a)
            elif corr == ABSEXP:
                dr = -a * rb
            else:
                raise ValueError(corr)
            dr[np.arange(i1 - i0), np.arange(i0, i1)] = 0.0
            quad += gamma[i0:i1].dot(dr).dot(gamma)
            trace += np.sum(rinv[i0:i1] * dr)
        grad[k] = (quad / sigma2n - trace) / math.log(10.0)
    return gp.likelihood, grad


# --------------------------------------------------------------------------
# Synthetic workload of SURVEY.md section 8(d) (shared by tests and bench.py's
# cpu_baseline leg; the product has its own copy in egobox_amd/workload.py).
# --------------------------------------------------------------------------
def lhs_classic(n, d, seed):
    """Classic LHS on [0,1]^d: (perm_j(i) + U)/n per column (crates/doe/src/lhs.rs:236-257
    is the reference's generator; bit parity with its Xoshiro256+ stream is not required)."""
    rng = np.random.default_rng(seed)
    x = np.empty((n, d))
    for j in range(d):
        x[:, j] = (rng.permutation(n) + rng.random(n)) / n
    return x


def griewank(x01):
    """crates/gp/benches/gp.rs:21-25 / python/egobox/tests/test_gpmix.py:10-21 on [-600,600]^d."""
    x = -600.0 + 1200.0 * np.asarray(x01)
    i = np.arange(1, x.shape[1] + 1)
    return (x * x).sum(axis=1) / 4000.0 - np.prod(np.cos(x / np.sqrt(i)), axis=1) + 1.0
