#!/usr/bin/env python3
"""What a TUNED fit through the builder costs around its likelihood evaluations (GpParams.fit, ThetaTuning::Full, 10 starts + the
initial guess on up to 12 workspaces; the reference: crates/gp/src/algorithm.rs:873-960): wall time of fit() against evaluations x
the lock-step evaluation rate, first and later fits of a shape, and closing the model."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

for n, d in ((1000, 4), (2048, 8), (4096, 8)):
    for rep in range(3):
        x, y = egx.workload.make_training_set(n, d, 11 + rep)
        p = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()).n_start(10).max_eval(25)
        t0 = time.perf_counter()
        g = p.fit(x, y)
        t1 = time.perf_counter()
        yq = g.predict(x[:100])
        t2 = time.perf_counter()
        g.close()
        t3 = time.perf_counter()
        print(f"n={n} d={d} fit {rep}: {1e3 * (t1 - t0):8.2f} ms for {g.n_evals} evaluations ({g.n_evals / (t1 - t0):7.1f}/s), "
              f"first predict {1e3 * (t2 - t1):6.2f} ms, close {1e3 * (t3 - t2):6.2f} ms", flush=True)
