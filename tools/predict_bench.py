#!/usr/bin/env python3
"""Side measurements on one GPU: K1 (correlation build) time of the four kernels at n = 16384 (d = 32, 64), and BASELINE
config 5's expert (n = 8192, d = 16): predict / predict_var / predict_valvar on 100 000 points.  One JSON line.
    python tools/predict_bench.py        (environment knobs EGX_CORR_TILE / EGX_TRSM_GROUP are read by the library)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

out = {"env": {k: v for k, v in os.environ.items() if k.startswith("EGX_")}}
n = 16384
for d in (32, 64):
    x, y = egx.workload.make_training_set(n, d, 42)
    th = egx.workload.default_theta(d)
    for corr, name in ((0, "sqexp"), (1, "absexp"), (2, "matern32"), (3, "matern52")):
        with egx.GpHandle(x, y, corr=corr) as h:
            h.likelihood(th)
            ts = []
            for _ in range(3):
                h.likelihood(th)
                ts.append(h.timings()["corr_build_ms"])
            out[f"k1_ms_{name}_d{d}"] = float(np.median(ts))
n5, d5, m5 = 8192, 16, 100000
x5, y5 = egx.workload.make_training_set(n5, d5, 7)
xq = np.random.default_rng(7).random((m5, d5))
with egx.GpHandle(x5, y5) as h:
    h.finalize(egx.workload.default_theta(d5))
    h.predict_valvar(xq[:4000])
    for name, fn in (("predict", h.predict), ("predict_var", h.predict_var), ("predict_valvar", h.predict_valvar)):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn(xq)
            ts.append(time.perf_counter() - t0)
        out[f"{name}_points_per_s"] = m5 / min(ts)
    out["predict_var_frac_of_fp64_peak"] = float(n5) * n5 * out["predict_var_points_per_s"] / 1e12 / 78.6
print(json.dumps(out), flush=True)
