"""Predict-side mixture-of-experts recombination around the GPU GP experts ("next").

Mirrors what egobox-moe does AFTER its clustering has chosen and trained the experts:
  GaussianMixture.predict_probas / predict        crates/moe/src/gaussian_mixture.rs:114-121, 231-283, 305-316
  GpMixture.predict_smooth / predict_var_smooth   crates/moe/src/algorithm.rs:411-423, 670-685, 789-809
  GpMixture.predict_hard  / predict_var_hard      crates/moe/src/algorithm.rs:879-935
  GpMixture.predict_(var_)gradients smooth / hard crates/moe/src/algorithm.rs:691-783, 942-1010;
  GaussianMixture.predict_probas_derivatives      crates/moe/src/gaussian_mixture.rs:127-170
Clustering itself (GMM fitting, expert selection by cross-validation) stays in egobox-moe.

Differences that matter on a GPU: the reference's hard recombination calls the expert ONCE PER ROW with a
1 x nx batch (each a full n^2 triangular solve for the variance); here queries are routed once and every
expert gets one batched call on its subset.  With several GPUs expert e lives on rank e mod G
(BASELINE config 5) and one all-reduce of the weighted vectors replaces the fold over experts.
With GPU experts the recombinations of values, variances AND their x-gradients, the responsibilities and their
derivatives run inside libegx_gp_hip.so (egx_moe_predict_valvar, egx_moe_predict_valvar_gradients,
egx_gmx_predict_probas(_derivatives)); the numpy forms below serve duck-typed experts and the CPU tests.
"""
from __future__ import annotations

import math

import numpy as np


class GaussianMixture:
    """Responsibilities of a fitted Gaussian mixture (weights (k,), means (k,nx), covariances (k,nx,nx))."""

    def __init__(self, weights, means, covariances, heaviside_factor=1.0):
        self.weights = np.asarray(weights, dtype=np.float64)
        self.means = np.atleast_2d(np.asarray(means, dtype=np.float64))
        self.covariances = np.asarray(covariances, dtype=np.float64)
        k, nx = self.means.shape
        if self.weights.shape != (k,) or self.covariances.shape != (k, nx, nx):
            raise ValueError("weights (k,), means (k,nx), covariances (k,nx,nx) expected")
        # precisions_chol[k] = (chol(cov_k)^-1)^T, gaussian_mixture.rs:182-205
        self.precisions_chol = np.empty((k, nx, nx))
        for i in range(k):
            c = np.linalg.cholesky(self.covariances[i])
            self.precisions_chol[i] = np.linalg.solve(c, np.eye(nx)).T
        self.set_heaviside_factor(heaviside_factor)

    def set_heaviside_factor(self, f):
        """gaussian_mixture.rs:105-110: refresh the log-determinants."""
        self.heaviside_factor = float(f)
        precs = self.precisions_chol * self.heaviside_factor ** -0.5
        self.log_det = np.log(np.einsum("kii->ki", precs)).sum(axis=1)
        return self

    @property
    def n_clusters(self):
        return self.means.shape[0]

    def _log_gaussian_prob(self, x):
        nx = self.means.shape[1]
        precs = self.precisions_chol * self.heaviside_factor ** -0.5
        q = np.empty((x.shape[0], self.n_clusters))
        for k in range(self.n_clusters):  # one (m,nx)x(nx,nx) product per cluster
            diff = (x - self.means[k]) @ precs[k]
            q[:, k] = np.einsum("ij,ij->i", diff, diff)
        return -0.5 * (q + nx * math.log(2.0 * math.pi)) + self.log_det

    def _log_resp(self, x):
        wlp = self._log_gaussian_prob(x) + np.log(self.weights)
        e = np.where(wlp <= -307.0, 0.0, np.exp(wlp))
        s = e.sum(axis=1)
        norm = np.where(np.abs(s) < np.finfo(float).eps, 0.0, np.log(np.where(s > 0, s, 1.0)))
        return wlp - norm[:, None]

    def predict_probas(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        if self.n_clusters == 1:
            return np.ones((x.shape[0], 1))
        return np.exp(self._log_resp(x))

    def predict_probas_device(self, x, device=-1):
        """The same responsibilities from the library (egx_gmx_predict_probas: one lane per point on the GPU); what the
        library-side recombination of `GpMixture` consumes."""
        from . import _lib as L
        lib = L.load()
        x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float64)))
        k, nx = self.means.shape
        out = np.empty((x.shape[0], k))
        L.check(lib.egx_gmx_predict_probas(int(device), L.dptr(np.ascontiguousarray(self.weights)),
                                           L.dptr(np.ascontiguousarray(self.means)),
                                           L.dptr(np.ascontiguousarray(self.precisions_chol)), k, nx,
                                           self.heaviside_factor, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def predict_probas_derivatives_device(self, x, device=-1):
        """d p_c / d x, (m, k, nx), from the library (egx_gmx_predict_probas_derivatives, one lane per point on the GPU)."""
        from . import _lib as L
        lib = L.load()
        x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float64)))
        k, nx = self.means.shape
        if 3 * nx + k > 320:  # (the kernel keeps one point per lane with 3 nx + k doubles of LDS scratch: beyond that, the host form)
            return self.predict_probas_derivatives(x)
        out = np.empty((x.shape[0], k, nx))
        L.check(lib.egx_gmx_predict_probas_derivatives(int(device), L.dptr(np.ascontiguousarray(self.weights)),
                                                       L.dptr(np.ascontiguousarray(self.means)),
                                                       L.dptr(np.ascontiguousarray(self.precisions_chol)), k, nx,
                                                       self.heaviside_factor, L.dptr(x), x.shape[0], L.dptr(out)))
        return out

    def predict(self, x):
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        return np.argmax(np.exp(self._log_resp(x)), axis=1)

    def pdfs(self, x):
        return np.exp(self._log_gaussian_prob(np.asarray(x, dtype=np.float64).reshape(1, -1))[0])

    def predict_probas_derivatives(self, x):
        """gaussian_mixture.rs:127-170, all points at once -> (m, k, nx): d p_i(x) / d x with p_i = u_i / v,
        u_i = w_i pdf_i(x), v = sum_i u_i."""
        x = np.atleast_2d(np.asarray(x, dtype=np.float64))
        u = self.weights * np.exp(self._log_gaussian_prob(x))                      # (m, k)
        v = u.sum(axis=1)                                                          # (m,)
        precs = np.einsum("kij,klj->kil", self.precisions_chol, self.precisions_chol) / self.heaviside_factor
        deriv = np.einsum("mkj,kjl->mkl", x[:, None, :] - self.means[None, :, :], precs)
        uprime = -deriv * u[:, :, None]                                            # (m, k, nx)
        vprime = uprime.sum(axis=1)                                                # (m, nx)
        return (uprime * v[:, None, None] - u[:, :, None] * vprime[:, None, :]) / (v * v)[:, None, None]


class GpMixture:
    """Experts + mixture, predict side only.  `experts[i]` is None for experts that live on another rank."""

    def __init__(self, experts, gmx, recombination="hard", rank=0, world=1, device=None, sweep=None):
        """`sweep`: the rank's `egobox_amd.Sweep` (its RCCL communicator carries the recombination's one all-gather
        inside the library, egx_moe_predict_valvar); without it a multi-rank mixture reduces through torch.distributed."""
        self.experts, self.gmx = list(experts), gmx
        self.sweep = sweep
        self.recombination = recombination.lower()
        if self.recombination not in ("hard", "smooth"):
            raise ValueError("recombination must be 'hard' or 'smooth'")
        if len(self.experts) != gmx.n_clusters:
            raise ValueError("one expert per cluster expected")
        self.rank, self.world, self.device = rank, world, device
        self.n_in_flight = 2

    @classmethod
    def fit_experts(cls, params, cluster_xs, cluster_ys, gmx, recombination="hard", **kw):
        """The expert loop of egobox-moe (crates/moe/src/algorithm.rs:167-177: one GP per cluster, fitted one after the other)
        with the experts of EQUAL training-set size fitted in lock-step: `params` is a `GpParams` with ThetaTuning.Fixed,
        cluster_xs[i] / cluster_ys[i] the training set of cluster i (the clustering itself is out of this package's scope).
        Clusters of one size go through `GpParams.fit_group` (one launch sequence for all of them), the others through
        `fit`; every expert is bit for bit what `fit` alone gives."""
        k = len(cluster_xs)
        experts = [None] * k
        by_shape = {}
        for i in range(k):
            by_shape.setdefault(np.asarray(cluster_xs[i]).shape, []).append(i)
        for shape, idx in by_shape.items():
            if len(idx) > 1:
                gps = params.fit_group(np.stack([np.asarray(cluster_xs[i], dtype=np.float64) for i in idx]),
                                       np.stack([np.asarray(cluster_ys[i], dtype=np.float64).reshape(shape[0]) for i in idx]))
                for i, g in zip(idx, gps):
                    experts[i] = g
            else:
                experts[idx[0]] = params.fit(cluster_xs[idx[0]], cluster_ys[idx[0]])
        return cls(experts, gmx, recombination, **kw)

    def _mine(self, i):
        return i % self.world == self.rank and self.experts[i] is not None

    def _allreduce(self, *arrays):
        if self.world == 1:
            return arrays
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.stack(arrays))
        if self.device is not None:
            t = t.to(self.device)
        dist.all_reduce(t)  # ncclAllReduce(sum) over xGMI; gloo in the CPU tests
        out = t.cpu().numpy()
        return tuple(out[i] for i in range(len(arrays)))

    def _library_handles(self):
        """The egx_gp* of this rank's experts when ALL of them are GPU handles (and the collective, if any, is the
        library's): then the recombination runs inside libegx_gp_hip.so (egx_moe_predict_valvar)."""
        if self.world > 1 and self.sweep is None:
            return None
        ids, hs = [], []
        for i, e in enumerate(self.experts):
            if not self._mine(i):
                continue
            h = getattr(getattr(e, "_h", None), "_h", None)
            if h is None or not h:
                return None
            ids.append(i)
            hs.append(h)
        return ids, hs

    def _predict_valvar_library(self, x, lib_handles, want_val, want_var):
        import ctypes as C
        from . import _lib as L
        lib = L.load()
        ids, hs = lib_handles
        m, d = x.shape
        k = len(self.experts)
        dev = getattr(self.gmx, "predict_probas_device", None)  # the library's kernel; a duck-typed mixture keeps its own
        probas = np.ascontiguousarray(dev(x) if dev is not None else self.gmx.predict_probas(x), dtype=np.float64)
        harr = (C.c_void_p * max(1, len(hs)))(*[h.value if hasattr(h, "value") else h for h in hs])
        iarr = np.asarray(ids, dtype=np.int32)
        val = np.empty(m) if want_val else None
        var = np.empty(m) if want_var else None
        L.check(lib.egx_moe_predict_valvar(self.sweep._h if self.sweep is not None else None, harr,
                                           iarr.ctypes.data_as(L.c_int32_p), len(hs), k, L.dptr(probas), L.dptr(x), m, d,
                                           1 if self.recombination == "smooth" else 0,
                                           L.dptr(val) if want_val else None, L.dptr(var) if want_var else None))
        return (val if want_val else np.zeros(m)), (var if want_var else np.zeros(m))

    def predict_valvar(self, x, want_val=True, want_var=True):
        x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float64)))
        m = x.shape[0]
        lib_handles = self._library_handles()
        if lib_handles is not None and m > 0:
            return self._predict_valvar_library(x, lib_handles, want_val, want_var)
        val, var = np.zeros(m), np.zeros(m)
        smooth = self.recombination == "smooth"
        if smooth:
            p = self.gmx.predict_probas(x)
        else:
            c = self.gmx.predict(x)

        def run(i):
            e = self.experts[i]
            idx = None if smooth else np.flatnonzero(c == i)
            if idx is not None and idx.size == 0:
                return i, idx, None, None
            xi = x if smooth else x[idx]
            if want_val and want_var:
                y, v = e.predict_valvar(xi)
            elif want_val:
                y, v = e.predict(xi), None
            else:
                y, v = None, e.predict_var(xi)
            return i, idx, y, v

        mine = [i for i in range(len(self.experts)) if self._mine(i)]
        # every expert owns a handle with its own HIP streams: a few in flight hide each other's launch gaps
        if len(mine) > 1 and self.n_in_flight > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(min(self.n_in_flight, len(mine))) as pool:
                results = list(pool.map(run, mine))
        else:
            results = [run(i) for i in mine]
        for i, idx, y, v in results:
            if smooth:
                if y is not None:
                    val += y * p[:, i]
                if v is not None:
                    var += v * p[:, i] * p[:, i]
            elif idx.size:
                if y is not None:
                    val[idx] = y
                if v is not None:
                    var[idx] = v
        val, var = self._allreduce(val, var)
        return val, var

    def predict(self, x):
        return self.predict_valvar(x, True, False)[0]

    def predict_var(self, x):
        return self.predict_valvar(x, False, True)[1]

    def _predict_valvar_gradients_library(self, x, lib_handles, want_val, want_var):
        import ctypes as C
        from . import _lib as L
        lib = L.load()
        ids, hs = lib_handles
        m, d = x.shape
        k = len(self.experts)
        smooth = self.recombination == "smooth"
        dev = getattr(self.gmx, "predict_probas_device", None)
        probas = np.ascontiguousarray(dev(x) if dev is not None else self.gmx.predict_probas(x), dtype=np.float64)
        dprobas = None
        if smooth and k > 1:
            ddev = getattr(self.gmx, "predict_probas_derivatives_device", None)
            dprobas = np.ascontiguousarray(ddev(x) if ddev is not None else self.gmx.predict_probas_derivatives(x),
                                           dtype=np.float64)
        harr = (C.c_void_p * max(1, len(hs)))(*[h.value if hasattr(h, "value") else h for h in hs])
        iarr = np.asarray(ids, dtype=np.int32)
        gy = np.empty((m, d)) if want_val else None
        gv = np.empty((m, d)) if want_var else None
        L.check(lib.egx_moe_predict_valvar_gradients(self.sweep._h if self.sweep is not None else None, harr,
                                                     iarr.ctypes.data_as(L.c_int32_p), len(hs), k, L.dptr(probas),
                                                     L.dptr(dprobas) if dprobas is not None else None, L.dptr(x), m, d,
                                                     1 if smooth else 0, L.dptr(gy) if want_val else None,
                                                     L.dptr(gv) if want_var else None))
        return (gy if want_val else np.zeros((m, d))), (gv if want_var else np.zeros((m, d)))

    def predict_valvar_gradients(self, x, want_val=True, want_var=True):
        """crates/moe/src/algorithm.rs:691-783 (smooth), :942-1010 (hard) -> ((m, nx), (m, nx)).
        smooth:  d mean = sum_i p_i grad y_i + p'_i y_i ;  d var = sum_i p_i^2 grad v_i + 2 p_i p'_i v_i.
        Every expert gets ONE batched call per quantity (the reference calls it once per row).  With GPU experts the
        recombination runs inside the library (egx_moe_predict_valvar_gradients, round 4); the numpy fold below serves
        duck-typed experts (the CPU tests)."""
        x = np.ascontiguousarray(np.atleast_2d(np.asarray(x, dtype=np.float64)))
        m, nx = x.shape
        lib_handles = self._library_handles()
        if lib_handles is not None and m > 0:
            return self._predict_valvar_gradients_library(x, lib_handles, want_val, want_var)
        gy, gv = np.zeros((m, nx)), np.zeros((m, nx))
        smooth = self.recombination == "smooth"
        if smooth:
            p = self.gmx.predict_probas(x)
            pp = self.gmx.predict_probas_derivatives(x) if self.gmx.n_clusters > 1 else np.zeros((m, 1, nx))
        else:
            c = self.gmx.predict(x)
        for i, e in enumerate(self.experts):
            if not self._mine(i):
                continue
            idx = None if smooth else np.flatnonzero(c == i)
            if idx is not None and idx.size == 0:
                continue
            xi = x if smooth else x[idx]
            if want_val:
                g = e.predict_gradients(xi)
                if smooth:
                    gy += g * p[:, i:i + 1] + pp[:, i, :] * e.predict(xi)[:, None]
                else:
                    gy[idx] = g
            if want_var:
                g = e.predict_var_gradients(xi)
                if smooth:
                    gv += g * (p[:, i:i + 1] ** 2) + 2.0 * p[:, i:i + 1] * pp[:, i, :] * e.predict_var(xi)[:, None]
                else:
                    gv[idx] = g
        gy, gv = self._allreduce(gy, gv)
        return gy, gv

    def predict_gradients(self, x):
        return self.predict_valvar_gradients(x, True, False)[0]

    def predict_var_gradients(self, x):
        return self.predict_valvar_gradients(x, False, True)[1]
