"""Mirror of the reference's Python surface for the hot path (python/src/gp_mix.rs:33-496):

    Gpx.builder(regr_spec, corr_spec, kpls_dim, n_clusters, recombination, theta_init, theta_bounds,
                n_start, max_eval, seed) -> GpMix ;  GpMix.fit(xt, yt) -> Gpx
    Gpx.predict / predict_var / thetas / variances / likelihoods / dims / training_data / save / load

Only what sits on the accelerated path is implemented: one cluster (n_clusters = 1), one regression and
one correlation spec.  Clustering, expert selection by cross-validation and the GMM recombination stay in
egobox-moe ("next").  `n_start = -1` is the reference's fixed-theta entry point
(gp_mix.rs:202-208) used for parity.
"""
from __future__ import annotations

import enum
import json

import numpy as np

from . import gp as G


class RegressionSpec(enum.IntFlag):  # python/src/types.rs RegressionSpec
    CONSTANT = 1
    LINEAR = 2
    QUADRATIC = 4
    ALL = 7


class CorrelationSpec(enum.IntFlag):  # python/src/types.rs CorrelationSpec
    SQUARED_EXPONENTIAL = 1
    ABSOLUTE_EXPONENTIAL = 2
    MATERN32 = 4
    MATERN52 = 8
    ALL = 15


class Recombination(enum.Enum):
    HARD = 0
    SMOOTH = 1


_REGR = {RegressionSpec.CONSTANT: G.ConstantMean, RegressionSpec.LINEAR: G.LinearMean,
         RegressionSpec.QUADRATIC: G.QuadraticMean}
_CORR = {CorrelationSpec.SQUARED_EXPONENTIAL: G.SquaredExponentialCorr,
         CorrelationSpec.ABSOLUTE_EXPONENTIAL: G.AbsoluteExponentialCorr,
         CorrelationSpec.MATERN32: G.Matern32Corr, CorrelationSpec.MATERN52: G.Matern52Corr}
_SURROGATE_NAME = {"ConstantMean": "Constant", "LinearMean": "Linear", "QuadraticMean": "Quadratic",
                   "SquaredExponential": "SquaredExponential", "AbsoluteExponential": "AbsoluteExponential",
                   "Matern32": "Matern32", "Matern52": "Matern52"}


def _single(spec, table, what):
    spec = type(next(iter(table)))(int(spec))
    hits = [v for k, v in table.items() if spec & k]
    if len(hits) != 1:
        raise NotImplementedError(
            f"{what}: expert selection among several specs is done by egobox-moe cross-validation "
            f"(crates/moe/src/algorithm.rs:209-347), outside the accelerated path; pass exactly one spec")
    return hits[0]()


def _nd(a):
    a = np.asarray(a, dtype=np.float64)
    return {"v": 1, "dim": list(a.shape), "data": a.ravel().tolist()}


def _from_nd(o):
    return np.asarray(o["data"], dtype=np.float64).reshape(o["dim"])


class GpMix:
    """python/src/gp_mix.rs:33-236."""

    def __init__(self, regr_spec=RegressionSpec.CONSTANT, corr_spec=CorrelationSpec.SQUARED_EXPONENTIAL,
                 kpls_dim=None, n_clusters=1, recombination=Recombination.HARD, theta_init=None,
                 theta_bounds=None, n_start=G.GP_OPTIM_N_START, max_eval=G.GP_COBYLA_MAX_EVAL, seed=None,
                 nugget=G.DEFAULT_NUGGET, device=-1):
        self.regr_spec, self.corr_spec = regr_spec, corr_spec
        self.kpls_dim, self.n_clusters, self.recombination = kpls_dim, n_clusters, recombination
        self.theta_init, self.theta_bounds = theta_init, theta_bounds
        self.n_start, self.max_eval, self.seed = n_start, max_eval, seed
        self.nugget, self.device = nugget, device

    def fit(self, xt, yt):
        if self.n_clusters != 1:
            raise NotImplementedError("clustering (n_clusters != 1) belongs to egobox-moe, outside the accelerated path")
        mean = _single(self.regr_spec, _REGR, "regr_spec")
        corr = _single(self.corr_spec, _CORR, "corr_spec")
        params = G.GpParams(mean, corr).nugget(self.nugget).device(self.device).max_eval(self.max_eval)
        tuning = G.ThetaTuning.default()  # gp_mix.rs:181-194
        if self.theta_init is not None:
            tuning = G.ThetaTuning.Full(self.theta_init, [G.ThetaTuning.DEFAULT_BOUNDS])
        if self.theta_bounds is not None:
            tuning = G.ThetaTuning.Full(tuning.init, [tuple(b) for b in self.theta_bounds])
        if self.n_start < 0:  # gp_mix.rs:202-208: no multistart, theta_init used as is
            tuning = G.ThetaTuning.Fixed(tuning.init)
            n_start = 0
        else:
            n_start = self.n_start
        params.theta_tuning(tuning).n_start(n_start).kpls_dim(self.kpls_dim)
        # `seed` feeds the mixture's clustering rng only (python/src/gp_mix.rs:177-181); the GP multistart LHS is always
        # seeded with 42 (crates/gp/src/optimization.rs:62), so params._seed keeps its default
        return Gpx([params.fit(xt, yt)], self)


class Gpx:
    """A trained Gaussian processes mixture (python/src/gp_mix.rs:240-496), single expert."""

    def __init__(self, experts, builder=None):
        self._experts = experts
        self._builder = builder

    @staticmethod
    def builder(regr_spec=RegressionSpec.CONSTANT, corr_spec=CorrelationSpec.SQUARED_EXPONENTIAL, kpls_dim=None,
                n_clusters=1, recombination=Recombination.HARD, theta_init=None, theta_bounds=None,
                n_start=G.GP_OPTIM_N_START, max_eval=G.GP_COBYLA_MAX_EVAL, seed=None, **kw):
        return GpMix(regr_spec, corr_spec, kpls_dim, n_clusters, recombination, theta_init, theta_bounds, n_start,
                     max_eval, seed, **kw)

    def predict(self, x):
        return self._experts[0].predict(x)

    def predict_var(self, x):
        return self._experts[0].predict_var(x)

    def predict_valvar(self, x):
        return self._experts[0].predict_valvar(x)

    def predict_gradients(self, x):
        """python/src/gp_mix.rs `predict_gradients`: (m, nx) derivatives of the mean."""
        return self._experts[0].predict_gradients(x)

    def predict_var_gradients(self, x):
        """python/src/gp_mix.rs `predict_var_gradients`: (m, nx) derivatives of the variance."""
        return self._experts[0].predict_var_gradients(x)

    def thetas(self):
        return np.stack([e.theta() for e in self._experts])

    def variances(self):
        return np.array([e.variance() for e in self._experts])

    def likelihoods(self):
        return np.array([e.likelihood() for e in self._experts])

    def dims(self):
        return self._experts[0].dims()

    def training_data(self):
        return self._experts[0].training_data

    def __str__(self):  # "Mixture[Hard](Linear_Matern52GP(mean=..., ...))"  doc/Gpx_Tutorial.ipynb cell 31
        parts = []
        for e in self._experts:
            m = _SURROGATE_NAME[str(e.params_._mean)]
            c = _SURROGATE_NAME[str(e.params_._corr)]
            parts.append(f"{m}_{c}{str(e)}")
        return "Mixture[Hard](" + ", ".join(parts) + ")"

    # ---- serde-compatible expert dump (schema: doc/Gpx_Tutorial.ipynb cell 31 output,
    #      crates/gp/src/algorithm.rs:41-60,165-192, crates/moe/src/surrogates.rs:108-248) ----
    def _expert_dict(self, e):
        ip = e.inner_params(with_chol=True)
        t = e.params_._theta_tuning
        x, y = e.training_data
        d = x.shape[1]
        tuning = {t.kind: {"init": _nd(t.init)}} if t.kind != "Fixed" else {"Fixed": _nd(t.init)}
        if t.kind != "Fixed":
            tuning[t.kind]["bounds"] = {"v": 1, "dim": [len(t.bounds)], "data": [list(b) for b in t.bounds]}
        if t.kind == "Partial":  # ThetaTuning::Partial { init, bounds, active } (crates/gp/src/parameters.rs:24-32)
            tuning[t.kind]["active"] = [int(a) for a in t.active]
        m = _SURROGATE_NAME[str(e.params_._mean)]
        c = _SURROGATE_NAME[str(e.params_._corr)]
        return {
            "type_fullgp": f"Gp{m}{c}Surrogate",
            "theta": _nd(ip["theta"]),
            "likelihood": ip["likelihood"],
            "inner_params": {"sigma2": ip["sigma2"], "beta": _nd(ip["beta"]), "gamma": _nd(ip["gamma"]),
                             "r_chol": _nd(ip["r_chol"]), "ft": _nd(ip["ft"]), "ft_qr_r": _nd(ip["ft_qr_r"])},
            "w_star": _nd(np.eye(d) if e.handle._w is None else e.handle._w),
            "xt_norm": {"data": _nd(ip["xt_norm"]), "mean": _nd(ip["x_mean"]), "std": _nd(ip["x_std"])},
            "yt_norm": {"data": _nd(ip["yt_norm"]), "mean": _nd(ip["y_mean"]), "std": _nd(ip["y_std"])},
            "training_data": [_nd(x), _nd(y)],
            "params": {"theta_tuning": tuning, "mean": str(e.params_._mean), "corr": str(e.params_._corr),
                       "kpls_dim": e.params_._kpls_dim, "n_start": e.params_._n_start,
                       "max_eval": e.params_._max_eval, "nugget": e.params_._nugget},
        }

    def to_dict(self):
        return {"recombination": "Hard", "experts": [self._expert_dict(e) for e in self._experts],
                "gp_type": "FullGp"}

    def __repr__(self):
        return json.dumps(self.to_dict())

    def save(self, filename):
        if not str(filename).endswith(".json"):
            raise NotImplementedError("only the JSON format is written (the reference's .bin is bincode)")
        with open(filename, "w") as f:
            json.dump(self.to_dict(), f)
        return True

    @staticmethod
    def load(filename, refit=False):
        """Rebuild the model on the GPU from a JSON dump (ours or the reference's serde schema,
        crates/moe/src/surrogates.rs:426-441).  The stored factor, gamma, beta, ft and ft_qr_r are uploaded as they
        are (`egx_gp_set_inner`); `refit=True` re-factors at the stored theta from the stored training data instead."""
        with open(filename) as f:
            obj = json.load(f)
        return Gpx.from_dict(obj, refit)

    @staticmethod
    def from_dict(obj, refit=False):
        experts = []
        for e in obj["experts"]:
            p = e["params"]
            mean = G.MEANS[p["mean"]]()
            corr = G.CORRS[p["corr"]]()
            x, y = _from_nd(e["training_data"][0]), _from_nd(e["training_data"][1])
            w = _from_nd(e["w_star"])
            theta = _from_nd(e["theta"])
            params = G.GpParams(mean, corr).nugget(p["nugget"]).theta_tuning(G.ThetaTuning.Fixed(theta))
            # KPLS is decided from the DATA: kpls_dim == nx gives a square, non-identity rotation (the reference only
            # rejects kpls_dim > nx, algorithm.rs:798-807), and r_chol / gamma were computed with the rotated kernel
            if p.get("kpls_dim") is not None or not np.array_equal(w, np.eye(w.shape[0], w.shape[1])):
                params.kpls_weights(w)
            if refit:
                experts.append(params.fit(x, y))
                continue
            h = G.GpHandle(x, y, mean=mean.code, corr=corr.code, nugget=p["nugget"],
                           w_star=params._kpls_weights)
            ip = e["inner_params"]
            h.set_inner(theta, e["likelihood"], ip["sigma2"], _from_nd(ip["beta"]), _from_nd(ip["gamma"]),
                        _from_nd(ip["r_chol"]), _from_nd(ip["ft"]), _from_nd(ip["ft_qr_r"]))
            experts.append(G.GaussianProcess(h, params, n_evals=0))
        return Gpx(experts)
