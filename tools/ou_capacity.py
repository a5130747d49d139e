#!/usr/bin/env python3
"""Capacity check: exact Ornstein-Uhlenbeck closed form (tests/test_gpu_parity.py::_ou_closed_form) at sizes that
fill a large part of the 288 GB of an MI355X.  usage: python tools/ou_capacity.py 98304 [131072 ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import egobox_amd as egx  # noqa: E402
from test_gpu_parity import _ou_closed_form  # noqa: E402

for n in [int(a) for a in sys.argv[1:]]:
    rng = np.random.default_rng(n)
    x = np.sort(rng.random(n)) + np.arange(n) * (0.5 / n)
    y = np.sin(7.0 * x) + 0.3 * np.cos(23.0 * x) + 0.05 * rng.standard_normal(n)
    theta = 40.0
    lk, s2, beta, mp = _ou_closed_form(x, y, theta)
    perm = rng.permutation(n)
    t0 = time.perf_counter()
    h = egx.GpHandle(x[perm].reshape(-1, 1), y[perm], corr=1, nugget=0.0)
    t1 = time.perf_counter()
    h.finalize([theta])
    t2 = time.perf_counter()
    lk_gpu, s2_gpu = h.fitted_scalars()
    tm = h.timings()
    print({"n": n, "matrix_GB": n * n * 8 / 1e9, "create_s": t1 - t0, "fit_s": t2 - t1,
           "potrf_ms": tm["potrf_ms"], "cholesky_tflops": tm["potrf_flops"] / tm["potrf_ms"] / 1e9,
           "likelihood_rel_err": abs(lk_gpu - lk) / abs(lk), "sigma2_rel_err": abs(s2_gpu - s2) / s2,
           "min_pivot_closed_form": mp}, flush=True)
    h.close()
