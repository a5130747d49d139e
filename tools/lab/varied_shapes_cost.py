#!/usr/bin/env python3
"""Models of DIFFERENT sizes one after the other, each created, fitted and dropped (a mixture's clusters, cross-validation folds,
candidate specifications: crates/moe/src/algorithm.rs:167-262): creation cost per model, lone and tuned (12 workspaces)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

x0, y0 = egx.workload.make_training_set(700, 4, 1)
with egx.GpHandle(x0, y0, corr=1) as h:
    h.finalize(np.full(4, 1.0))
egx.trim()
d = 4
tot_c = tot_f = 0.0
for i, n in enumerate((900, 1300, 1700, 2100, 2600, 3100, 3600, 4100, 5000, 6000)):
    x, y = egx.workload.make_training_set(n, d, n)
    t0 = time.perf_counter()
    h = egx.GpHandle(x, y, corr=1, n_workspaces=1)
    t1 = time.perf_counter()
    h.finalize(np.full(d, 1.0))
    t2 = time.perf_counter()
    h.close()
    t3 = time.perf_counter()
    tot_c += t1 - t0 + t3 - t2
    tot_f += t2 - t1
    print(f"lone n={n}: create {1e3 * (t1 - t0):6.2f} ms, fit {1e3 * (t2 - t1):6.2f} ms, close {1e3 * (t3 - t2):5.2f} ms", flush=True)
print(f"ten lone models of ten sizes: create + close {1e3 * tot_c:.1f} ms, fits {1e3 * tot_f:.1f} ms", flush=True)
egx.trim()
tot = 0.0
for n in (500, 800, 1100, 1500, 1900, 2400):
    x, y = egx.workload.make_training_set(n, d, n + 1)
    p = egx.GaussianProcess.params(egx.ConstantMean(), egx.AbsoluteExponentialCorr()).n_start(10).max_eval(25)
    t0 = time.perf_counter()
    g = p.fit(x, y)
    t1 = time.perf_counter()
    g.close()
    tot += t1 - t0
    print(f"tuned n={n}: fit() {1e3 * (t1 - t0):7.2f} ms for {g.n_evals} evaluations", flush=True)
print(f"six tuned models of six sizes: {1e3 * tot:.1f} ms", flush=True)
