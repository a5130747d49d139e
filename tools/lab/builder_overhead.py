#!/usr/bin/env python3
"""Python builder against the handle it wraps, at the reference's everyday sizes: GaussianProcess.params(...).fit(x, y) (fixed theta
and tuned), predict of one point / of 1000 points through the model object."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402


def t(fn, reps=30):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


for n, d in ((100, 2), (256, 4), (1000, 8), (4096, 8)):
    x, y = egx.workload.make_training_set(n, d, 3)
    th = np.full(d, 1.0)
    pf = egx.GaussianProcess.params(egx.ConstantMean(), egx.AbsoluteExponentialCorr()).theta_tuning(egx.ThetaTuning.Fixed(th))
    pt = egx.GaussianProcess.params(egx.ConstantMean(), egx.AbsoluteExponentialCorr()).n_start(10).max_eval(25)

    def fit_fixed():
        g = pf.fit(x, y)
        g.close()

    def fit_tuned():
        g = pt.fit(x, y)
        g.close()

    def handle_fixed():
        h = egx.GpHandle(x, y, corr=1)
        h.finalize(th)
        h.close()

    g = pf.fit(x, y)
    xq1 = x[:1] + 0.01
    xq = np.random.default_rng(0).random((1000, d))
    print(f"n={n} d={d}: builder fixed-theta fit {t(fit_fixed):.3f} ms (handle create + finalize + close {t(handle_fixed):.3f}), "
          f"builder tuned fit {t(fit_tuned, 5):.2f} ms, model.predict 1 point {t(lambda: g.predict(xq1)):.3f} ms "
          f"(handle {t(lambda: g.handle.predict(xq1)):.3f}), predict_var 1 point {t(lambda: g.predict_var(xq1)):.3f} ms, "
          f"predict 1000 points {t(lambda: g.predict(xq)):.3f} ms", flush=True)
    g.close()
x, y = egx.workload.make_training_set(256, 4, 3)
pf = egx.GaussianProcess.params(egx.ConstantMean(), egx.AbsoluteExponentialCorr()).theta_tuning(egx.ThetaTuning.Fixed(np.full(4, 1.0)))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    g = pf.fit(x, y)
    g.close()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
