#!/usr/bin/env python3
"""rocprofv3 target: 300 likelihood evaluations at n = 256 and n = 512, d = 8 (kernel mix of the EGO-sized problems)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402
from egobox_amd import workload  # noqa: E402

for n in (256, 512):
    x, y = workload.make_training_set(n, 8, 1)
    h = egx.GpHandle(x, y, corr=3)
    th = np.full(8, 1.0)
    for _ in range(5):
        h.likelihood(th)
    t0 = time.perf_counter()
    for _ in range(300):
        h.likelihood(th)
    print(n, (time.perf_counter() - t0) / 300 * 1e3, "ms")
    h.close()
