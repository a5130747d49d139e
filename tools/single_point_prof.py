#!/usr/bin/env python3
"""rocprofv3 target: 200 single-point predict_gradients / predict_var_gradients calls at n = 2048 (kernel mix of the small path)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402
from egobox_amd import workload  # noqa: E402

x, y = workload.make_training_set(2048, 8, 1)
h = egx.GpHandle(x, y, corr=3)
h.finalize(np.full(8, 1.0))
xq = np.random.default_rng(0).random((1, 8))
for name, fn in (("predict_gradients", h.predict_gradients), ("predict_var_gradients", h.predict_var_gradients),
                 ("predict", h.predict), ("predict_var", h.predict_var)):
    for _ in range(5):
        fn(xq)
    t0 = time.perf_counter()
    for _ in range(200):
        fn(xq)
    print(name, (time.perf_counter() - t0) / 200 * 1e3, "ms")
