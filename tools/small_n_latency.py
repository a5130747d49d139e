#!/usr/bin/env python3
"""Latency of the path at the small sizes EGO actually uses (n = 100 .. 4000): fixed-theta fit, one likelihood
evaluation, predict / predict_var / gradients of ONE point."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402
from egobox_amd import workload  # noqa: E402


def t(fn, reps=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


for n in (128, 256, 512, 1024, 2048, 4096):
    d = 8
    x, y = workload.make_training_set(n, d, 1)
    th = np.full(d, 1.0)
    h = egx.GpHandle(x, y, corr=3)
    xq = np.random.default_rng(0).random((1, d))
    out = {"n": n, "d": d, "likelihood_ms": t(lambda: h.likelihood(th)), "fit_fixed_ms": t(lambda: h.finalize(th))}
    out["predict_1_ms"] = t(lambda: h.predict(xq))
    out["predict_var_1_ms"] = t(lambda: h.predict_var(xq))
    out["predict_valvar_1_ms"] = t(lambda: h.predict_valvar(xq))
    out["predict_gradients_1_ms"] = t(lambda: h.predict_gradients(xq))
    out["predict_var_gradients_1_ms"] = t(lambda: h.predict_var_gradients(xq))
    tm = h.timings()
    out["stage_ms"] = {k: round(tm[k], 3) for k in ("corr_build_ms", "potrf_ms", "solve_ms", "host_ms", "total_ms")}
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}), flush=True)
    h.close()
