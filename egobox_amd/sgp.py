"""Sparse Gaussian process (FITC / VFE) on the GPU -- mirror of the reference's two surfaces for it:

    Rust    SparseGaussianProcess::params(Inducings::Located(z) | Randomized(n)).sparse_method(..).noise_variance(..)
            .theta_init(..).theta_bounds(..).n_start(..).max_eval(..).seed(..).fit(x, y)
            -> predict / predict_var / theta / variance / noise_variance / likelihood / inducings
            (crates/gp/src/sparse_parameters.rs, sparse_algorithm.rs:145-300, 422-650)
    Python  SparseGpx.builder(corr_spec, theta_init, theta_bounds, n_start, nz | z, method, seed).fit(xt, yt)
            (python/src/sparse_gp_mix.rs)

The sparse GP works on raw x / y (no normalisation) with a zero trend.  Inducings.Randomized(n) draws rows of x with
numpy's generator: the reference shuffles with Rust's Xoshiro256Plus, so the drawn points (and everything fitted from
them) are not comparable digit by digit; with Located(z) the likelihood at given parameters is.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from . import gp as G
from .multistart import prepare_multistart

FITC, VFE = 0, 1
DEFAULT_NOISE_INIT = 1e-2                                   # ParamTuning::default, sparse_parameters.rs:33-39
DEFAULT_NOISE_BOUNDS = (100.0 * np.finfo(float).eps, 1e10)
DEFAULT_THETA_BOUNDS = (1e-2, 1e2)                          # sparse_parameters.rs:160-163


class Inducings:
    def __init__(self, kind, value):
        self.kind, self.value = kind, value

    @classmethod
    def Randomized(cls, n):
        return cls("Randomized", int(n))

    @classmethod
    def Located(cls, z):
        return cls("Located", np.atleast_2d(np.asarray(z, dtype=np.float64)))


class ParamTuning:
    """Noise variance: Fixed(c) or Optimized{init, bounds} (sparse_parameters.rs:13-40)."""

    def __init__(self, kind, init, bounds=None):
        self.kind, self.init, self.bounds = kind, float(init), bounds

    @classmethod
    def Fixed(cls, c):
        return cls("Fixed", c)

    @classmethod
    def Optimized(cls, init=DEFAULT_NOISE_INIT, bounds=DEFAULT_NOISE_BOUNDS):
        return cls("Optimized", init, tuple(bounds))


class SgpHandle:
    """Thin owner of an `egx_sgp*`."""

    def __init__(self, x, y, z, corr=0, method=FITC, nugget=G.DEFAULT_NUGGET, device=-1):
        self._lib = L.load()
        x = L.as_f64(x)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        y = L.as_f64(np.asarray(y, dtype=np.float64).reshape(-1))
        z = L.as_f64(z)
        if z.ndim == 1:
            z = z.reshape(-1, x.shape[1])
        if x.ndim != 2 or z.ndim != 2 or z.shape[1] != x.shape[1] or y.shape[0] != x.shape[0]:
            raise L.InvalidValueError(L.ERR_INVALID_VALUE, "x (n, d), y (n), z (nz, d) expected")
        cfg = L.SgpConfig()
        self._lib.egx_sgp_config_default(C.byref(cfg))
        cfg.corr, cfg.method, cfg.nugget, cfg.device = int(corr), int(method), float(nugget), int(device)
        h = C.c_void_p()
        L.check(self._lib.egx_sgp_create(C.byref(cfg), L.dptr(x), L.dptr(y), x.shape[0], x.shape[1], L.dptr(z),
                                         z.shape[0], C.byref(h)))
        self._h = h
        self.n, self.d, self.nz = x.shape[0], x.shape[1], z.shape[0]
        self.z = z.copy()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.egx_sgp_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def likelihood(self, theta, sigma2, noise):
        th = L.as_f64(np.atleast_1d(theta), 1)
        lk, st = C.c_double(), C.c_int32()
        L.check(self._lib.egx_sgp_likelihood(self._h, L.dptr(th), th.size, float(sigma2), float(noise), C.byref(lk),
                                             C.byref(st)))
        return lk.value, st.value

    def finalize(self, theta, sigma2, noise):
        th = L.as_f64(np.atleast_1d(theta), 1)
        L.check(self._lib.egx_sgp_finalize(self._h, L.dptr(th), th.size, float(sigma2), float(noise)))

    def fit(self, params0s, lo, hi, estimate_noise, noise_fixed, max_eval):
        p0 = L.as_f64(params0s, 2)
        lo, hi = L.as_f64(lo, 1), L.as_f64(hi, 1)
        ne = C.c_int64()
        L.check(self._lib.egx_sgp_fit(self._h, L.dptr(p0), p0.shape[0], L.dptr(lo), L.dptr(hi), int(bool(estimate_noise)),
                                      float(noise_fixed), int(max_eval), C.byref(ne)))
        return ne.value

    def _q(self, x):
        x = L.as_f64(x)
        if x.ndim == 1:
            x = x.reshape(-1, self.d)
        if x.ndim != 2 or x.shape[1] != self.d:
            raise L.InvalidValueError(L.ERR_INVALID_VALUE, f"query points must be (m, {self.d}), got {x.shape}")
        return np.ascontiguousarray(x)

    def predict(self, x):
        x = self._q(x)
        out = np.empty(x.shape[0])
        L.check(self._lib.egx_sgp_predict(self._h, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def predict_var(self, x):
        x = self._q(x)
        out = np.empty(x.shape[0])
        L.check(self._lib.egx_sgp_predict_var(self._h, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def state(self, with_inv=False):
        th, vec = np.empty(self.d), np.empty(self.nz)
        s2, nv, lk = C.c_double(), C.c_double(), C.c_double()
        inv = np.empty((self.nz, self.nz)) if with_inv else None
        L.check(self._lib.egx_sgp_get_state(self._h, L.dptr(th), C.byref(s2), C.byref(nv), C.byref(lk), L.dptr(vec),
                                            L.dptr(inv) if with_inv else None))
        out = dict(theta=th, sigma2=s2.value, noise=nv.value, likelihood=lk.value, w_vec=vec)
        if with_inv:
            out["w_inv"] = inv
        return out


class SgpParams:
    """Builder, crates/gp/src/sparse_parameters.rs:150-300."""

    def __init__(self, corr, inducings):
        self._corr = corr
        self._inducings = inducings
        self._method = FITC
        self._noise = ParamTuning.Optimized()
        self._theta_tuning = G.ThetaTuning.Full([G.ThetaTuning.DEFAULT_INIT], [DEFAULT_THETA_BOUNDS])
        self._n_start, self._max_eval = G.GP_OPTIM_N_START, G.GP_COBYLA_MAX_EVAL
        self._nugget, self._seed, self._device = G.DEFAULT_NUGGET, None, -1

    def sparse_method(self, m):
        self._method = {"fitc": FITC, "vfe": VFE}[m.lower()] if isinstance(m, str) else int(m)
        return self

    def noise_variance(self, tuning):
        self._noise = tuning
        return self

    def theta_init(self, init):
        t = self._theta_tuning
        self._theta_tuning = G.ThetaTuning(t.kind, init, t.bounds)
        return self

    def theta_bounds(self, bounds):
        t = self._theta_tuning
        self._theta_tuning = G.ThetaTuning(t.kind, t.init, [tuple(b) for b in bounds])
        return self

    def theta_tuning(self, t):
        self._theta_tuning = t
        return self

    def n_start(self, n):
        self._n_start = int(n)
        return self

    def max_eval(self, n):
        self._max_eval = int(n)
        return self

    def nugget(self, v):
        self._nugget = float(v)
        return self

    def seed(self, s):
        self._seed = s
        return self

    def device(self, dev):
        self._device = int(dev)
        return self

    def fit(self, x, y):
        """SgpValidParams::fit, sparse_algorithm.rs:422-650."""
        x = np.asarray(x, dtype=np.float64)
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        if self._inducings.kind == "Located":
            z = self._inducings.value
        else:  # make_inducings :833-848 (numpy shuffle instead of Xoshiro256Plus)
            rng = np.random.default_rng(self._seed)
            z = x[rng.permutation(x.shape[0])[:min(self._inducings.value, x.shape[0])]].copy()
        h = SgpHandle(x, y, z, corr=self._corr.code, method=self._method, nugget=self._nugget, device=self._device)
        d = h.d
        t = self._theta_tuning
        init = np.asarray(t.init, dtype=np.float64).reshape(-1)
        if init.size not in (1, d):
            h.close()
            raise L.InvalidValueError(L.ERR_INVALID_VALUE,
                                      f"Initial guess for theta should be either 1-dim or dim of xtrain, got {init.size}")
        theta0 = np.full(d, init[0]) if init.size == 1 else init
        sigma2_0 = float(np.std(y, ddof=1) ** 2)                      # :503-505
        est = self._noise.kind == "Optimized"
        if t.kind == "Fixed":                                          # :479: bounds collapse onto the value
            tb = [(v, v) for v in theta0]
        else:
            tb = list(t.bounds) * d if len(t.bounds) == 1 else list(t.bounds)
        params0 = np.concatenate([theta0, [sigma2_0], [self._noise.init] if est else []])
        bounds = tb + [(1e-12, 9.0 * sigma2_0)] + ([tuple(self._noise.bounds)] if est else [])  # :580-592
        for i, (lo, hi) in enumerate(bounds):   # keep every start inside its box
            params0[i] = min(max(params0[i], lo), hi)
        starts_log10, _ = prepare_multistart(self._n_start, params0, bounds, seed=42 if self._seed is None else self._seed)
        n_evals = h.fit(10.0 ** starts_log10, [b[0] for b in bounds], [b[1] for b in bounds], est,
                        self._noise.init, self._max_eval)
        sgp = SparseGaussianProcess(h, self, n_evals)
        sgp._xy = (x, y)
        return sgp


class SparseGaussianProcess:
    """crates/gp/src/sparse_algorithm.rs:145-300 (predict side)."""

    def __init__(self, handle, params, n_evals=0):
        self._h, self.params_, self.n_evals = handle, params, n_evals
        self._xy = None

    @staticmethod
    def params(inducings, corr=None):
        return SgpParams(corr if corr is not None else G.SquaredExponentialCorr(), inducings)

    def predict(self, x):
        return self._h.predict(x)

    def predict_var(self, x):
        return self._h.predict_var(x)

    def _central_diff(self, fn, x):
        """sparse_algorithm.rs:298-336: the reference differentiates the sparse predictions NUMERICALLY
        (finitediff's central_diff, step sqrt(f64::EPSILON)); here all 2 nx shifted batches go to the GPU at once."""
        x = self._h._q(x)
        m, nx = x.shape
        h = float(np.sqrt(np.finfo(float).eps))
        shifted = np.repeat(x[None, :, :], 2 * nx, axis=0)
        for k in range(nx):
            shifted[2 * k, :, k] += h
            shifted[2 * k + 1, :, k] -= h
        v = fn(shifted.reshape(-1, nx)).reshape(2 * nx, m)
        return ((v[0::2] - v[1::2]) / (2.0 * h)).T.copy()

    def predict_gradients(self, x):
        return self._central_diff(self._h.predict, x)

    def predict_var_gradients(self, x):
        return self._central_diff(self._h.predict_var, x)

    def theta(self):
        return self._h.state()["theta"]

    def variance(self):
        return self._h.state()["sigma2"]

    def noise_variance(self):
        return self._h.state()["noise"]

    def likelihood(self):
        return self._h.state()["likelihood"]

    def inducings(self):
        return self._h.z

    def dims(self):
        return self._h.d, 1

    def woodbury(self):
        s = self._h.state(with_inv=True)
        return s["w_vec"], s["w_inv"]

    def close(self):
        self._h.close()

    def __str__(self):  # "SGP(corr=..., theta=..., variance=..., noise variance=..., likelihood=...)" :200-208
        s = self._h.state()
        return (f"SGP(corr={self.params_._corr}, theta={s['theta'].tolist()}, variance={s['sigma2']}, "
                f"noise variance={s['noise']}, likelihood={s['likelihood']})")


class SparseMethod:
    FITC = FITC
    VFE = VFE


class SparseGpMix:
    """python/src/sparse_gp_mix.rs: SparseGpx.builder(...) -> SparseGpMix, .fit(xt, yt) -> SparseGpx (single expert)."""

    def __init__(self, corr_spec=1, theta_init=None, theta_bounds=None, kpls_dim=None, n_start=G.GP_OPTIM_N_START, nz=None,
                 z=None, method=FITC, seed=None, max_eval=G.GP_COBYLA_MAX_EVAL, device=-1):
        if kpls_dim is not None:
            raise NotImplementedError("KPLS rotations come from linfa-pls (outside the accelerated path)")
        if nz is None and z is None:
            raise ValueError("either nz or z has to be specified")  # sparse_gp_mix.rs builder check
        self.corr_spec, self.theta_init, self.theta_bounds = int(corr_spec), theta_init, theta_bounds
        self.n_start, self.nz, self.z, self.method, self.seed = n_start, nz, z, method, seed
        self.max_eval, self.device = max_eval, device

    def fit(self, xt, yt):
        yt = np.asarray(yt, dtype=np.float64)
        if yt.ndim == 2 and yt.shape[1] != 1:
            raise ValueError("sparse GP mixture handles a single output")  # test_sgp_multi_outputs_exception
        corrs = {1: G.SquaredExponentialCorr, 2: G.AbsoluteExponentialCorr, 4: G.Matern32Corr, 8: G.Matern52Corr}
        if self.corr_spec not in corrs:
            raise NotImplementedError("pass exactly one correlation spec (expert selection belongs to egobox-moe)")
        ind = Inducings.Located(self.z) if self.z is not None else Inducings.Randomized(self.nz)
        p = SgpParams(corrs[self.corr_spec](), ind).sparse_method(self.method).n_start(self.n_start) \
            .max_eval(self.max_eval).seed(self.seed).device(self.device)
        if self.theta_init is not None:
            p.theta_init(self.theta_init)
        if self.theta_bounds is not None:
            p.theta_bounds(self.theta_bounds)
        return SparseGpx(p.fit(xt, yt.reshape(-1)))


class SparseGpx:
    def __init__(self, sgp):
        self._sgp = sgp

    @staticmethod
    def builder(**kw):
        return SparseGpMix(**kw)

    def predict(self, x):
        return self._sgp.predict(x)

    def predict_var(self, x):
        return self._sgp.predict_var(x)

    def predict_gradients(self, x):
        return self._sgp.predict_gradients(x)

    def predict_var_gradients(self, x):
        return self._sgp.predict_var_gradients(x)

    def thetas(self):
        return self._sgp.theta()[None, :]

    def variances(self):
        return np.array([self._sgp.variance()])

    def likelihoods(self):
        return np.array([self._sgp.likelihood()])

    def __str__(self):
        return f"Mixture[Smooth(1)]({self._sgp.params_._corr}{self._sgp})"

    # ---- JSON dump in the field layout of the reference's serde structs (SparseGaussianProcess :145-168, WoodburyData
    #      :32-36; typetag "type_sgp" crates/moe/src/surrogates.rs:101): theta, sigma2, noise, likelihood, w_star,
    #      inducings, w_data{vec, inv}, training_data, corr, method
    def to_dict(self):
        g = self._sgp
        h = g._h
        st = h.state(with_inv=True)

        def nd(a):
            a = np.asarray(a, dtype=np.float64)
            return {"v": 1, "dim": list(a.shape), "data": a.ravel().tolist()}

        corr = str(g.params_._corr)
        expert = {"type_sgp": f"Sgp{corr}Surrogate", "corr": corr, "method": "Fitc" if g.params_._method == FITC else "Vfe",
                  "theta": nd(st["theta"]), "sigma2": st["sigma2"], "noise": st["noise"], "likelihood": st["likelihood"],
                  "w_star": nd(np.eye(h.d)), "inducings": nd(h.z),
                  "w_data": {"vec": nd(st["w_vec"].reshape(-1, 1)), "inv": nd(st["w_inv"])},
                  "training_data": [nd(g._xy[0]), nd(g._xy[1])], "nugget": g.params_._nugget}
        return {"recombination": {"Smooth": 1.0}, "experts": [expert], "gp_type": "SparseGp"}

    def save(self, filename):
        import json
        if not str(filename).endswith(".json"):
            raise NotImplementedError("only the JSON format is written (the reference's .bin is bincode)")
        with open(filename, "w") as f:
            json.dump(self.to_dict(), f)
        return True

    @staticmethod
    def load(filename):
        """Rebuild on the GPU from a JSON dump: the stored (theta, sigma2, noise, inducings) are re-evaluated (one
        likelihood evaluation, milliseconds) and the stored likelihood is checked against the recomputed one."""
        import json
        with open(filename) as f:
            e = json.load(f)["experts"][0]

        def arr(o):
            return np.asarray(o["data"], dtype=np.float64).reshape(o["dim"])

        corr = G.CORRS[e["corr"]]()
        x, y = arr(e["training_data"][0]), arr(e["training_data"][1])
        z, theta = arr(e["inducings"]), arr(e["theta"])
        method = FITC if e["method"] == "Fitc" else VFE
        params = SgpParams(corr, Inducings.Located(z)).sparse_method(method).nugget(e.get("nugget", G.DEFAULT_NUGGET))
        h = SgpHandle(x, y, z, corr=corr.code, method=method, nugget=params._nugget)
        h.finalize(theta, e["sigma2"], e["noise"])
        lk = h.state()["likelihood"]
        if not np.isclose(lk, e["likelihood"], rtol=1e-6, atol=1e-9):
            h.close()
            raise L.EgxError(L.ERR_LIKELIHOOD, f"stored likelihood {e['likelihood']} != recomputed {lk}")
        sgp = SparseGaussianProcess(h, params, n_evals=1)
        sgp._xy = (x, y)
        return SparseGpx(sgp)
