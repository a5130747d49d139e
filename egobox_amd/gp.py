"""Host-side mirror of egobox-gp's builder / fit / predict API on top of the HIP library.

Same names, argument meaning and error behaviour as the reference for the hot path:

    GaussianProcess.params(mean, corr) / Kriging.params()          crates/gp/src/algorithm.rs:200-207, 244-249
      .theta_init .theta_bounds .theta_tuning .kpls_dim .n_start
      .max_eval .nugget                                             crates/gp/src/parameters.rs:167-273
      .fit(x, y) -> GaussianProcess                                 crates/gp/src/algorithm.rs:785-980
    GaussianProcess.predict / predict_var / predict_valvar          crates/gp/src/algorithm.rs:253-307
      .theta() .variance() .likelihood() .dims() .kpls_dim()        crates/gp/src/algorithm.rs:413-439

All numerics run on the GPU through `GpHandle` (ctypes over include/egx_gp.h); nothing here
computes a correlation, a factorisation or a solve.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .multistart import prepare_multistart

#: crates/gp/src/lib.rs:  GP_OPTIM_N_START = 10, GP_COBYLA_MIN_EVAL = 25, GP_COBYLA_MAX_EVAL = 1000
GP_OPTIM_N_START = 10
GP_COBYLA_MIN_EVAL = 25
GP_COBYLA_MAX_EVAL = 1000
DEFAULT_NUGGET = 100.0 * np.finfo(np.float64).eps  # parameters.rs:118


# ---- model markers (crates/gp/src/mean_models.rs, correlation_models.rs) ---------------------
class _Named:
    code = -1
    name = ""

    def __str__(self):
        return self.name

    def __repr__(self):
        return f"{type(self).__name__}()"

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self))


class ConstantMean(_Named):
    code, name = 0, "ConstantMean"


class LinearMean(_Named):
    code, name = 1, "LinearMean"


class QuadraticMean(_Named):
    code, name = 2, "QuadraticMean"


class SquaredExponentialCorr(_Named):
    code, name = 0, "SquaredExponential"


class AbsoluteExponentialCorr(_Named):
    code, name = 1, "AbsoluteExponential"


class Matern32Corr(_Named):
    code, name = 2, "Matern32"


class Matern52Corr(_Named):
    code, name = 3, "Matern52"


MEANS = {c.name: c for c in (ConstantMean, LinearMean, QuadraticMean)}
CORRS = {c.name: c for c in (SquaredExponentialCorr, AbsoluteExponentialCorr, Matern32Corr, Matern52Corr)}


class ThetaTuning:
    """crates/gp/src/parameters.rs:14-78."""
    DEFAULT_INIT = 1e-1
    DEFAULT_BOUNDS = (1e-2, 1e1)

    def __init__(self, kind, init, bounds=None, active=None):
        self.kind = kind
        self.init = np.atleast_1d(np.asarray(init, dtype=np.float64))
        self.bounds = None if bounds is None else [tuple(map(float, b)) for b in bounds]
        self.active = None if active is None else list(active)

    @classmethod
    def Fixed(cls, init):
        return cls("Fixed", init)

    @classmethod
    def Full(cls, init=None, bounds=None):
        return cls("Full", [cls.DEFAULT_INIT] if init is None else init,
                   [cls.DEFAULT_BOUNDS] if bounds is None else bounds)

    @classmethod
    def Partial(cls, init, bounds, active):
        return cls("Partial", init, bounds, active)

    @classmethod
    def default(cls):
        return cls.Full()

    def __repr__(self):
        return f"ThetaTuning.{self.kind}(init={self.init.tolist()}, bounds={self.bounds}, active={self.active})"


# ---- low level handle --------------------------------------------------------------------------
class GpHandle:
    """One training set resident on one GPU (opaque `egx_gp*`)."""

    def __init__(self, x, y, mean=0, corr=0, nugget=DEFAULT_NUGGET, device=-1, n_workspaces=1, w_star=None):
        lib = L.load()
        x = L.as_f64(x)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        if x.ndim != 2:
            raise L.InvalidValueError(L.ERR_INVALID_VALUE, "Training input has to be an [nsamples, nx] array")
        y = L.as_f64(y)
        if y.ndim == 2 and y.shape[1] == 1:
            y = y[:, 0]
        if y.ndim != 1:
            raise L.InvalidValueError(L.ERR_INVALID_VALUE, "Training output has to be one dimensional")
        if y.shape[0] != x.shape[0]:
            raise L.InvalidValueError(L.ERR_INVALID_VALUE,
                                      f"x has {x.shape[0]} rows but y has {y.shape[0]} (ragged training set)")
        y = np.ascontiguousarray(y)
        cfg = L.GpConfig()
        lib.egx_gp_config_default(C.byref(cfg))
        cfg.corr, cfg.mean, cfg.nugget, cfg.device, cfg.n_workspaces = int(corr), int(mean), float(nugget), int(device), int(n_workspaces)
        self._w = None
        if w_star is not None:
            self._w = L.as_f64(w_star, 2)
            if self._w.shape[0] != x.shape[1]:
                raise L.InvalidValueError(L.ERR_INVALID_VALUE, "w_star must be (nx, kpls_dim)")
            cfg.w_star = L.dptr(self._w)
            cfg.kpls_dim = self._w.shape[1]
        self._h = C.c_void_p()
        self._lib = lib
        L.check(lib.egx_gp_create(C.byref(cfg), L.dptr(x), L.dptr(y), x.shape[0], x.shape[1], C.byref(self._h)))
        n, d, p, h = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        L.check(lib.egx_gp_dims(self._h, C.byref(n), C.byref(d), C.byref(p), C.byref(h)))
        self.n, self.d, self.p, self.h = n.value, d.value, p.value, h.value

    @classmethod
    def create_group(cls, xs, ys, mean=0, corr=0, nugget=DEFAULT_NUGGET, device=-1):
        """egx_gp_create_group: k models of ONE shape (xs: k x n x d, ys: k x n) whose matrices share a slab -- the experts of a
        mixture (crates/moe/src/algorithm.rs:167-177), an optimiser's objective and constraint surrogates -- so that
        `finalize_multi` / `likelihood_multi` factor them in lock-step.  Returns k ordinary handles."""
        lib = L.load()
        xs = np.ascontiguousarray(L.as_f64(xs))
        ys = np.ascontiguousarray(L.as_f64(ys))
        if xs.ndim != 3 or ys.ndim != 2 or ys.shape != xs.shape[:2]:
            raise L.InvalidValueError(L.ERR_INVALID_VALUE, "create_group needs xs (k, n, d) and ys (k, n)")
        k, n, d = xs.shape
        cfg = L.GpConfig()
        lib.egx_gp_config_default(C.byref(cfg))
        cfg.corr, cfg.mean, cfg.nugget, cfg.device, cfg.n_workspaces = int(corr), int(mean), float(nugget), int(device), 1
        raw = (C.c_void_p * k)()
        L.check(lib.egx_gp_create_group(C.byref(cfg), L.dptr(xs), L.dptr(ys), n, d, k, raw))
        out = []
        for j in range(k):
            self = cls.__new__(cls)
            self._lib, self._h, self._w = lib, C.c_void_p(raw[j]), None
            self.n, self.d = n, d
            nn, dd, p, h = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
            L.check(lib.egx_gp_dims(self._h, C.byref(nn), C.byref(dd), C.byref(p), C.byref(h)))
            self.p, self.h = p.value, h.value
            out.append(self)
        return out

    @classmethod
    def _borrow(cls, raw, owner):
        """A view of an egx_gp* owned by something else (the replica inside an `egx_sweep`): every method works, closing it
        does nothing; `owner` is kept alive as long as the view is."""
        self = cls.__new__(cls)
        self._lib = L.load()
        self._h = C.c_void_p(raw if isinstance(raw, int) else raw.value)
        self._w = None
        self._owner = owner
        n, d, p, h = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self._lib.egx_gp_dims(self._h, C.byref(n), C.byref(d), C.byref(p), C.byref(h)))
        self.n, self.d, self.p, self.h = n.value, d.value, p.value, h.value
        return self

    @property
    def training_data(self):
        """(x, y) as given to the fit: the handle's own copy (the fitted model owns its training data,
        algorithm.rs:969-978), fetched from the library when asked for instead of being duplicated in Python."""
        x, y = np.empty((self.n, self.d)), np.empty(self.n)
        L.check(self._lib.egx_gp_get_training_data(self._h, L.dptr(x), L.dptr(y)))
        return x, y

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if getattr(self, "_owner", None) is None:
                self._lib.egx_gp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- likelihood
    def set_lockstep(self, width):
        """Candidates of `likelihood_batch` factored in lock-step by one launch sequence (egx_gp_set_lockstep);
        0 = the library's default, 1 = one stream set per candidate."""
        L.check(self._lib.egx_gp_set_lockstep(self._h, int(width)))
        return self._lib.egx_gp_get_lockstep(self._h)

    def schedule(self):
        """egx_gp_get_schedule: how this handle factors (decided per handle, so all its evaluations agree bit for bit)."""
        out = (C.c_int32 * 7)()
        L.check(self._lib.egx_gp_get_schedule(self._h, out, 7))
        keys = ("left_looking", "left_looking_rider", "pipelined_chain", "whole_factorisation_launch", "panels_per_group", "lockstep", "flow")
        return dict(zip(keys, [int(v) for v in out]))

    def shrink(self, n_keep=1):
        """egx_gp_shrink: keep the first n_keep workspaces (the fitted factor lives in workspace 0), free the rest."""
        L.check(self._lib.egx_gp_shrink(self._h, int(n_keep)))

    def likelihood(self, theta):
        theta = L.as_f64(np.atleast_1d(theta), 1)
        lk, st = C.c_double(), C.c_int32()
        L.check(self._lib.egx_gp_likelihood(self._h, L.dptr(theta), theta.size, C.byref(lk), C.byref(st)))
        return lk.value, st.value

    def likelihood_batch(self, thetas):
        thetas = L.as_f64(thetas, 2)
        k = thetas.shape[0]
        lk = np.empty(k)
        st = np.empty(k, dtype=np.int32)
        L.check(self._lib.egx_gp_likelihood_batch(self._h, L.dptr(thetas), k, thetas.shape[1], L.dptr(lk),
                                                  st.ctypes.data_as(L.c_int32_p)))
        return lk, st

    def likelihood_grad(self, theta):
        theta = L.as_f64(np.atleast_1d(theta), 1)
        lk, st = C.c_double(), C.c_int32()
        g = np.zeros(self.h)
        L.check(self._lib.egx_gp_likelihood_grad(self._h, L.dptr(theta), theta.size, C.byref(lk), L.dptr(g),
                                                 C.byref(st)))
        return lk.value, g, st.value

    def likelihood_grad_batch(self, thetas):
        """(likelihoods (k), gradients (k x h), statuses (k)) of the rows of thetas: egx_gp_likelihood_grad_batch."""
        thetas = L.as_f64(thetas, 2)
        k = thetas.shape[0]
        lk = np.empty(k)
        g = np.zeros((k, self.h))
        st = np.empty(k, dtype=np.int32)
        L.check(self._lib.egx_gp_likelihood_grad_batch(self._h, L.dptr(thetas), k, thetas.shape[1], L.dptr(lk), L.dptr(g),
                                                       st.ctypes.data_as(L.c_int32_p)))
        return lk, g, st

    # -- fit
    def finalize(self, theta):
        theta = L.as_f64(np.atleast_1d(theta), 1)
        L.check(self._lib.egx_gp_finalize(self._h, L.dptr(theta), theta.size))

    def fit(self, theta0s, lo, hi, max_eval=GP_COBYLA_MAX_EVAL):
        theta0s = L.as_f64(theta0s, 2)
        lo = L.as_f64(np.atleast_1d(lo), 1)
        hi = L.as_f64(np.atleast_1d(hi), 1)
        ne = C.c_int64()
        L.check(self._lib.egx_gp_fit(self._h, L.dptr(theta0s), theta0s.shape[0], L.dptr(lo), L.dptr(hi), lo.size,
                                     int(max_eval), C.byref(ne)))
        return ne.value

    def fit_partial(self, theta_init, active, theta0s, lo, hi, max_eval=GP_COBYLA_MAX_EVAL):
        theta_init = L.as_f64(theta_init, 1)
        act = np.ascontiguousarray(active, dtype=np.int64)
        theta0s = L.as_f64(theta0s, 2)
        lo = L.as_f64(np.atleast_1d(lo), 1)
        hi = L.as_f64(np.atleast_1d(hi), 1)
        ne = C.c_int64()
        L.check(self._lib.egx_gp_fit_partial(self._h, L.dptr(theta_init), act.ctypes.data_as(L.c_int64_p), act.size,
                                             L.dptr(theta0s), theta0s.shape[0], L.dptr(lo), L.dptr(hi), lo.size,
                                             int(max_eval), C.byref(ne)))
        return ne.value

    def fit_lbfgs(self, theta0s, lo, hi, max_iter=50):
        theta0s = L.as_f64(theta0s, 2)
        lo = L.as_f64(np.atleast_1d(lo), 1)
        hi = L.as_f64(np.atleast_1d(hi), 1)
        ne = C.c_int64()
        L.check(self._lib.egx_gp_fit_lbfgs(self._h, L.dptr(theta0s), theta0s.shape[0], L.dptr(lo), L.dptr(hi), lo.size,
                                           int(max_iter), C.byref(ne)))
        return ne.value

    # -- predict
    def _q(self, x):
        x = L.as_f64(x)
        if x.ndim == 1:
            x = x.reshape(-1, self.d) if self.d > 1 else x.reshape(-1, 1)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise L.InvalidValueError(L.ERR_INVALID_VALUE, f"query points must be (m, {self.d}), got {x.shape}")
        return np.ascontiguousarray(x)

    def predict(self, x):
        x = self._q(x)
        out = np.empty(x.shape[0])
        L.check(self._lib.egx_gp_predict(self._h, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def predict_var(self, x):
        x = self._q(x)
        out = np.empty(x.shape[0])
        L.check(self._lib.egx_gp_predict_var(self._h, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def predict_valvar(self, x):
        x = self._q(x)
        y = np.empty(x.shape[0])
        v = np.empty(x.shape[0])
        L.check(self._lib.egx_gp_predict_valvar(self._h, L.dptr(x), x.shape[0], L.dptr(y), L.dptr(v)))
        return y, v

    # -- x-gradients (m, d): d prediction / d x in original units
    def predict_gradients(self, x):
        x = self._q(x)
        out = np.empty((x.shape[0], self.d))
        L.check(self._lib.egx_gp_predict_gradients(self._h, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def predict_var_gradients(self, x):
        x = self._q(x)
        out = np.empty((x.shape[0], self.d))
        L.check(self._lib.egx_gp_predict_var_gradients(self._h, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def predict_valvar_gradients(self, x):
        x = self._q(x)
        gy = np.empty((x.shape[0], self.d))
        gv = np.empty((x.shape[0], self.d))
        L.check(self._lib.egx_gp_predict_valvar_gradients(self._h, L.dptr(x), x.shape[0], L.dptr(gy), L.dptr(gv)))
        return gy, gv

    # -- state
    def inner(self, with_chol=False):
        n, d, p, h = self.n, self.d, self.p, self.h
        out = dict(theta=np.empty(h), likelihood=np.empty(1), sigma2=np.empty(1), beta=np.empty((p, 1)),
                   gamma=np.empty((n, 1)), ft=np.empty((n, p)), ft_qr_r=np.empty((p, p)), x_mean=np.empty(d),
                   x_std=np.empty(d), y_mean=np.empty(1), y_std=np.empty(1), xt_norm=np.empty((n, d)),
                   yt_norm=np.empty((n, 1)))
        if with_chol:
            out["r_chol"] = np.empty((n, n))
        view = L.InnerView()
        for k, a in out.items():
            setattr(view, k, L.dptr(a))
        L.check(self._lib.egx_gp_get_inner(self._h, C.byref(view)))
        out["likelihood"] = float(out["likelihood"][0])
        out["sigma2"] = float(out["sigma2"][0])
        return out

    def set_inner(self, theta, likelihood, sigma2, beta, gamma, r_chol, ft, ft_qr_r):
        """Install a fitted state produced elsewhere (deserialised model) without re-factoring."""
        n, p, h = self.n, self.p, self.h
        arrs = dict(theta=np.broadcast_to(L.as_f64(np.atleast_1d(theta)).ravel(), (h,)).copy(),
                    likelihood=np.array([float(likelihood)]), sigma2=np.array([float(sigma2)]),
                    beta=L.as_f64(beta).reshape(p).copy(), gamma=L.as_f64(gamma).reshape(n).copy(),
                    r_chol=L.as_f64(r_chol).reshape(n, n).copy(), ft=L.as_f64(ft).reshape(n, p).copy(),
                    ft_qr_r=L.as_f64(ft_qr_r).reshape(p, p).copy())
        view = L.InnerView()
        for k, a in arrs.items():
            setattr(view, k, L.dptr(a))
        L.check(self._lib.egx_gp_set_inner(self._h, C.byref(view)))

    def fitted_scalars(self):
        """(likelihood, sigma2) of the resident fit without downloading any array."""
        lk, s2 = np.empty(1), np.empty(1)
        view = L.InnerView()
        view.likelihood, view.sigma2 = L.dptr(lk), L.dptr(s2)
        L.check(self._lib.egx_gp_get_inner(self._h, C.byref(view)))
        return float(lk[0]), float(s2[0])

    def timings(self):
        t = L.Timings()
        L.check(self._lib.egx_gp_last_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in L.Timings._fields_}


# ---- kernel-level entry points -------------------------------------------------------------------
def corr_matrix(corr, xnorm, theta, nugget=DEFAULT_NUGGET):
    """CorrelationModel::value fused with DiffMatrix + the R scatter loop (egx_corr_matrix)."""
    xnorm = L.as_f64(xnorm, 2)
    n, d = xnorm.shape
    theta = np.ascontiguousarray(np.broadcast_to(L.as_f64(np.atleast_1d(theta), 1), (d,)))
    r = np.empty((n, n))
    L.check(L.load().egx_corr_matrix(int(corr), L.dptr(xnorm), n, d, L.dptr(theta), float(nugget), L.dptr(r)))
    return r


def cross_corr(corr, xq_norm, xt_norm, theta):
    xq_norm = L.as_f64(xq_norm, 2)
    xt_norm = L.as_f64(xt_norm, 2)
    m, d = xq_norm.shape
    n = xt_norm.shape[0]
    theta = np.ascontiguousarray(np.broadcast_to(L.as_f64(np.atleast_1d(theta), 1), (d,)))
    r = np.empty((m, n))
    L.check(L.load().egx_cross_corr(int(corr), L.dptr(xq_norm), m, L.dptr(xt_norm), n, d, L.dptr(theta), L.dptr(r)))
    return r


def potrf(a):
    """Lower Cholesky factor on the GPU (egx_potrf). Returns (L, info)."""
    a = np.array(a, dtype=np.float64, order="C", copy=True)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise L.InvalidValueError(L.ERR_INVALID_VALUE, "square matrix expected")
    info = C.c_int32()
    L.check(L.load().egx_potrf(L.dptr(a), a.shape[0], C.byref(info)))
    return a, info.value


def normalize(x):
    x = L.as_f64(x, 2)
    n, d = x.shape
    xn, mean, std = np.empty((n, d)), np.empty(d), np.empty(d)
    L.check(L.load().egx_normalize(L.dptr(x), n, d, L.dptr(xn), L.dptr(mean), L.dptr(std)))
    return xn, mean, std


def regression_basis(mean, x):
    x = L.as_f64(x, 2)
    n, d = x.shape
    p = L.load().egx_regression_ncols(int(mean), d)
    if p < 0:
        raise L.InvalidValueError(L.ERR_INVALID_VALUE, "unknown regression model")
    f = np.empty((n, p))
    L.check(L.load().egx_regression_basis(int(mean), L.dptr(x), n, d, L.dptr(f)))
    return f


def mfma_probe():
    e = C.c_double()
    L.check(L.load().egx_mfma_probe(C.byref(e)))
    return e.value


def trim():
    """Free the device resources destroyed handles left in the library's pool (egx_trim); returns the bytes freed."""
    return int(L.load().egx_trim())


def _handle_array(handles):
    arr = (C.c_void_p * len(handles))()
    for j, h in enumerate(handles):
        arr[j] = getattr(h, "handle", h)._h
    return arr


def finalize_multi(handles, thetas):
    """egx_gp_finalize_multi: `fit` at fixed theta of several models at once (row j of thetas for handles[j]); members of one
    group (GpHandle.create_group) are factored in lock-step, each exactly as `finalize` would factor it alone."""
    thetas = L.as_f64(thetas, 2)
    if thetas.shape[0] != len(handles):
        raise L.InvalidValueError(L.ERR_INVALID_VALUE, "one theta row per model")
    L.check(L.load().egx_gp_finalize_multi(_handle_array(handles), len(handles), L.dptr(thetas), thetas.shape[1]))


def likelihood_multi(handles, thetas):
    """egx_gp_likelihood_multi: (likelihoods, statuses) of several models, row j of thetas for handles[j]."""
    thetas = L.as_f64(thetas, 2)
    if thetas.shape[0] != len(handles):
        raise L.InvalidValueError(L.ERR_INVALID_VALUE, "one theta row per model")
    lk = np.empty(len(handles))
    st = np.empty(len(handles), dtype=np.int32)
    L.check(L.load().egx_gp_likelihood_multi(_handle_array(handles), len(handles), L.dptr(thetas), thetas.shape[1], L.dptr(lk),
                                              st.ctypes.data_as(C.POINTER(C.c_int32))))
    return lk, st


def fit_multi(handles, theta0s, lo, hi, max_eval=GP_COBYLA_MAX_EVAL):
    """egx_gp_fit_multi: ThetaTuning::Full for several models at once (theta0s: k x n_starts x h in linear theta units); the
    COBYLA machines of all models advance in lock-step, members of one group are evaluated as one launch sequence.  Returns the
    evaluations per model; every model ends fitted as `fit` on a one-workspace handle would leave it."""
    theta0s = np.ascontiguousarray(L.as_f64(theta0s))
    if theta0s.ndim != 3 or theta0s.shape[0] != len(handles):
        raise L.InvalidValueError(L.ERR_INVALID_VALUE, "fit_multi needs theta0s (k, n_starts, h)")
    lo, hi = L.as_f64(np.atleast_1d(lo), 1), L.as_f64(np.atleast_1d(hi), 1)
    ne = np.zeros(len(handles), dtype=np.int64)
    L.check(L.load().egx_gp_fit_multi(_handle_array(handles), len(handles), L.dptr(theta0s), theta0s.shape[1], L.dptr(lo), L.dptr(hi),
                                      lo.size, int(max_eval), ne.ctypes.data_as(L.c_int64_p)))
    return ne


def set_tuning(knob, value):
    """egx_set_tuning: one of the factorisation's scheduling knobs by name; returns the previous value."""
    old = C.c_int32()
    L.check(L.load().egx_set_tuning(knob.encode(), int(value), C.byref(old)))
    return old.value


def pool_stats():
    """(cached bytes, hits, misses) of the resource pool behind egx_gp_create / egx_gp_destroy."""
    b, h, m = C.c_int64(), C.c_int64(), C.c_int64()
    L.load().egx_pool_stats(C.byref(b), C.byref(h), C.byref(m))
    return {"cached_bytes": b.value, "hits": h.value, "misses": m.value}


def chain_stats():
    """egx_chain_stats: evaluations whose chain launch ran into its wait bound, and how many of them were run again by separate
    launches (process-wide, since start-up)."""
    a, r = C.c_int64(), C.c_int64()
    L.load().egx_chain_stats(C.byref(a), C.byref(r))
    return {"aborted": a.value, "retried": r.value}


# ---- builder (crates/gp/src/parameters.rs:93-313) ---------------------------------------------------
class GpParams:
    def __init__(self, mean=None, corr=None):
        self._mean = mean if mean is not None else ConstantMean()
        self._corr = corr if corr is not None else SquaredExponentialCorr()
        self._theta_tuning = ThetaTuning.default()
        self._kpls_dim = None
        self._kpls_weights = None
        self._n_start = GP_OPTIM_N_START
        self._max_eval = GP_COBYLA_MAX_EVAL
        self._nugget = DEFAULT_NUGGET
        self._device = -1
        self._n_workspaces = None  # concurrent likelihood evaluations during a tuned fit; None = all starts (see fit)
        self._optimizer = "cobyla"  # the reference's optimiser (optimization.rs:122-169); "lbfgs" uses the new gradient
        self._seed = 42  # optimization.rs:62: multistart LHS is seeded with 42

    # setters return self, like the Rust builder
    def mean(self, mean):
        self._mean = mean
        return self

    def corr(self, corr):
        self._corr = corr
        return self

    def kpls_dim(self, kpls_dim):
        self._kpls_dim = kpls_dim
        return self

    def kpls_weights(self, w_star):
        """Extension: supply the PLS rotations (nx, kpls_dim) directly (the reference computes them with
        linfa-pls, algorithm.rs:843-855, which is outside the accelerated path)."""
        self._kpls_weights = None if w_star is None else np.asarray(w_star, dtype=np.float64)
        if w_star is not None:
            self._kpls_dim = self._kpls_weights.shape[1]
        return self

    def theta_init(self, theta_init):  # parameters.rs:193-212
        t = self._theta_tuning
        if t.kind == "Fixed":
            self._theta_tuning = ThetaTuning.Fixed(theta_init)
        else:
            self._theta_tuning = ThetaTuning.Full(theta_init, t.bounds)
        return self

    def theta_bounds(self, theta_bounds):  # parameters.rs:217-234 (no-op when Fixed)
        t = self._theta_tuning
        if t.kind != "Fixed":
            self._theta_tuning = ThetaTuning.Full(t.init, theta_bounds)
        return self

    def theta_tuning(self, theta_tuning):
        self._theta_tuning = theta_tuning
        return self

    def n_start(self, n_start):
        self._n_start = int(n_start)
        return self

    def max_eval(self, max_eval):  # parameters.rs:251-254
        self._max_eval = max(GP_COBYLA_MIN_EVAL, int(max_eval))
        return self

    def nugget(self, nugget):
        self._nugget = float(nugget)
        return self

    def device(self, device):
        self._device = int(device)
        return self

    def optimizer(self, name):
        """"cobyla" (default: Powell's COBYLA as the reference uses it, csrc/cobyla.h), or the extension "lbfgs"
        (projected L-BFGS on log10 theta driven by the new likelihood gradient)."""
        if name not in ("cobyla", "lbfgs"):
            raise ValueError("optimizer must be 'cobyla' or 'lbfgs'")
        self._optimizer = name
        return self

    def n_workspaces(self, n):
        """Extension: correlation-matrix workspaces = multistart optimisations run concurrently on the GPU
        (the reference runs its starts on a rayon pool, algorithm.rs:928-945)."""
        self._n_workspaces = max(1, int(n))
        return self

    def check(self):  # ParamGuard::check_ref, parameters.rs:287-308
        d = self._kpls_dim
        if d is not None:
            if d == 0:
                raise L.InvalidValueError(L.ERR_INVALID_VALUE, "`kpls_dim` canot be 0!")
            th = self._theta_tuning.init
            if th.size > 1 and d > th.size:
                raise L.InvalidValueError(
                    L.ERR_INVALID_VALUE,
                    f"Dimension reduction ({d}) should be smaller than expected training input size "
                    f"infered from given initial theta length ({th.size})")
        return self

    def fit(self, x, y):
        """GpValidParams::fit, crates/gp/src/algorithm.rs:791-980."""
        self.check()
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        nx = x.shape[1] if x.ndim == 2 else 0
        w = None
        if self._kpls_dim is not None:
            if self._kpls_dim > nx:  # algorithm.rs:798-807
                raise L.InvalidValueError(
                    L.ERR_INVALID_VALUE,
                    f"Dimension reduction {self._kpls_dim} should be smaller than actual training input dimensions {nx}")
            if self._kpls_weights is None:  # algorithm.rs:843-855: rotations of a PLS regression of y on x
                from .kpls import pls_rotations
                w = pls_rotations(x, y, self._kpls_dim)
            else:
                w = self._kpls_weights
        if self._theta_tuning.kind == "Fixed":
            nws = 1
        elif self._n_workspaces is not None:
            nws = min(self._n_workspaces, self._n_start + 1)
        else:
            # every start of the multistart gets a workspace (the reference runs them on a rayon pool, algorithm.rs:928-945):
            # COBYLA advances all starts in lock-step, so a round's trial points are ONE likelihood batch, factored in
            # lock-step (one slot below n_pad 14336, groups of four beyond) -- bounded by 12 workspaces and 16 GiB of correlation matrices
            n_pad = -(-x.shape[0] // 128) * 128
            nws = max(1, min(self._n_start + 1, 12, int((16 << 30) // max(1, 8 * n_pad * (n_pad + 128)))))
        h = GpHandle(x, y, mean=self._mean.code, corr=self._corr.code, nugget=self._nugget, device=self._device,
                     n_workspaces=max(1, nws), w_star=w)
        t = self._theta_tuning
        dim = h.h
        if t.init.size not in (1, dim):  # algorithm.rs:829-838 (a panic in the reference)
            h.close()
            raise L.InvalidValueError(
                L.ERR_INVALID_VALUE,
                f"Initial guess for theta should be either 1-dim or dim of xtrain (w_star.ncols()), got {t.init.size}")
        if t.kind == "Fixed":
            h.finalize(t.init)
            n_evals = 1
        elif t.kind == "Full":
            theta0 = np.full(dim, t.init[0]) if t.init.size == 1 else t.init
            b = t.bounds
            if len(b) not in (1, dim):  # algorithm.rs:901-912
                h.close()
                raise L.InvalidValueError(
                    L.ERR_INVALID_VALUE,
                    f"Bounds for theta should be either 1-dim or dim of xtrain ({dim}), got {len(b)}")
            b = b * dim if len(b) == 1 else b
            starts_log10, _ = prepare_multistart(self._n_start, theta0, b, seed=self._seed)
            if self._optimizer == "lbfgs":
                n_evals = h.fit_lbfgs(10.0 ** starts_log10, [lo for lo, _ in b], [hi for _, hi in b],
                                      max(5, min(100, self._max_eval // 4)))
            else:
                n_evals = h.fit(10.0 ** starts_log10, [lo for lo, _ in b], [hi for _, hi in b], self._max_eval)
        else:  # ThetaTuning::Partial, algorithm.rs:822-826, 873-960: only the active components move
            theta0 = np.full(dim, t.init[0]) if t.init.size == 1 else np.array(t.init, dtype=np.float64)
            b = t.bounds
            if len(b) not in (1, dim):
                h.close()
                raise L.InvalidValueError(
                    L.ERR_INVALID_VALUE,
                    f"Bounds for theta should be either 1-dim or dim of xtrain ({dim}), got {len(b)}")
            b = b * dim if len(b) == 1 else b
            active = sorted(set(int(i) for i in t.active))
            if not active or active[0] < 0 or active[-1] >= dim:
                h.close()
                raise L.InvalidValueError(L.ERR_INVALID_VALUE, f"active components must be indices in [0, {dim})")
            ab = [b[i] for i in active]
            starts_log10, _ = prepare_multistart(self._n_start, theta0[active], ab, seed=self._seed)
            n_evals = h.fit_partial(theta0, active, 10.0 ** starts_log10, [lo for lo, _ in ab],
                                    [hi for _, hi in ab], self._max_eval)
        n_pad = -(-x.shape[0] // 128) * 128
        if nws > 2 and (nws - 2) * 8 * n_pad * (n_pad + 128) >= (4 << 30):
            # the multistart's workspaces go back (egx_gp_shrink) when they hold 4 GiB or more: the resident model keeps the
            # factor's workspace and one more, so that likelihood evaluations on the fitted model do not un-fit it.  (Below
            # that it is not worth it: tearing down a workspace's streams and ~400 events costs ~20 ms -- 0.18 s for nine of
            # them, against 0.41 s for a whole tuned fit at n = 4096 and 45 ms at n = 1024.)
            h.shrink(2)
        return GaussianProcess(h, self, n_evals)

    def fit_group(self, xs, ys):
        """`fit` for k training sets of ONE shape at once (xs: k x n x d, ys: k x n; ThetaTuning Fixed or Full, no KPLS): what
        the expert loop of egobox-moe does one model after the other (crates/moe/src/algorithm.rs:167-177).  The models are
        created into one group of slabs (egx_gp_create_group) and factored in lock-step (egx_gp_finalize_multi); each is
        bit for bit the model `fit` returns for its training set.  Returns k GaussianProcess objects."""
        self.check()
        t = self._theta_tuning
        if t.kind not in ("Fixed", "Full") or self._kpls_dim is not None or (t.kind == "Full" and self._optimizer == "lbfgs"):
            raise L.InvalidValueError(L.ERR_INVALID_VALUE, "fit_group: ThetaTuning Fixed or Full (COBYLA), no dimension reduction")
        xs = np.asarray(xs, dtype=np.float64)
        ys = np.asarray(ys, dtype=np.float64)
        if ys.ndim == 3 and ys.shape[2] == 1:
            ys = ys[:, :, 0]
        hs = GpHandle.create_group(xs, ys, mean=self._mean.code, corr=self._corr.code, nugget=self._nugget, device=self._device)
        dim = hs[0].h
        if t.init.size not in (1, dim):
            for h in hs:
                h.close()
            raise L.InvalidValueError(
                L.ERR_INVALID_VALUE,
                f"Initial guess for theta should be either 1-dim or dim of xtrain (w_star.ncols()), got {t.init.size}")
        n_evals = [1] * len(hs)
        try:
            if t.kind == "Fixed":
                finalize_multi(hs, np.tile(np.atleast_1d(t.init), (len(hs), 1)) if t.init.size == dim
                               else np.full((len(hs), dim), float(t.init[0])))
            else:
                # ThetaTuning::Full (round 6): every model's multistart -- the same starts for all of them, as the reference's
                # per-expert fits draw theirs from one fixed seed (optimization.rs:49-66) -- through egx_gp_fit_multi
                theta0 = np.full(dim, t.init[0]) if t.init.size == 1 else t.init
                b = t.bounds
                if len(b) not in (1, dim):  # algorithm.rs:901-912
                    raise L.InvalidValueError(
                        L.ERR_INVALID_VALUE, f"Bounds for theta should be either 1-dim or dim of xtrain ({dim}), got {len(b)}")
                b = b * dim if len(b) == 1 else b
                starts_log10, _ = prepare_multistart(self._n_start, theta0, b, seed=self._seed)
                n_evals = [int(v) for v in fit_multi(hs, np.tile(10.0 ** starts_log10, (len(hs), 1, 1)), [lo for lo, _ in b],
                                                     [hi for _, hi in b], self._max_eval)]
        except Exception:
            for h in hs:
                h.close()
            raise
        return [GaussianProcess(h, self, ne) for h, ne in zip(hs, n_evals)]


class GaussianProcess:
    """Fitted model (crates/gp/src/algorithm.rs:174-192); state lives on the GPU."""

    def __init__(self, handle, params, n_evals=1):
        self._h = handle
        self.params_ = params
        self.n_evals = n_evals
        self._inner = None

    @staticmethod
    def params(mean=None, corr=None):
        return GpParams(mean, corr)

    def predict(self, x):
        return self._h.predict(x)

    def predict_var(self, x):
        return self._h.predict_var(x)

    def predict_valvar(self, x):
        return self._h.predict_valvar(x)

    def predict_gradients(self, x):
        """algorithm.rs:510-519 -> (m, nx)."""
        return self._h.predict_gradients(x)

    def predict_var_gradients(self, x):
        """algorithm.rs:702-709 -> (m, nx)."""
        return self._h.predict_var_gradients(x)

    def predict_valvar_gradients(self, x):
        """algorithm.rs:711-727 -> ((m, nx), (m, nx))."""
        return self._h.predict_valvar_gradients(x)

    def inner_params(self, with_chol=False):
        if self._inner is None or (with_chol and "r_chol" not in self._inner):
            self._inner = self._h.inner(with_chol)
        return self._inner

    def theta(self):
        return self.inner_params()["theta"]

    def variance(self):
        return self.inner_params()["sigma2"]

    def likelihood(self):
        return self.inner_params()["likelihood"]

    def dims(self):
        return (self._h.d, 1)

    def kpls_dim(self):
        return self._h.h if self._h._w is not None else None

    @property
    def training_data(self):
        return self._h.training_data

    @property
    def handle(self):
        return self._h

    def __str__(self):  # algorithm.rs:226-240
        th = ", ".join(repr(float(v)) for v in self.theta())
        return (f"GP(mean={self.params_._mean}, corr={self.params_._corr}, theta=[{th}], "
                f"variance={self.variance()!r}, likelihood={self.likelihood()!r})")

    def close(self):
        self._h.close()


class Kriging:
    """Kriging = GP with constant mean and squared exponential correlation (algorithm.rs:198-207)."""

    @staticmethod
    def params():
        return GpParams(ConstantMean(), SquaredExponentialCorr())
