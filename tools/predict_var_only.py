#!/usr/bin/env python3
"""One expert of BASELINE config 5 (n = 8192, d = 16): fit, then predict_var on 65 536 points twice -- the command behind
profiles/r03_predict_var_kernel_stats.txt (rocprofv3 --kernel-trace --stats)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

x, y = egx.workload.make_training_set(8192, 16, 7)
xq = np.random.default_rng(7).random((65536, 16))
with egx.GpHandle(x, y) as h:
    h.finalize(egx.workload.default_theta(16))
    h.predict_var(xq[:16384])
    import time
    t0 = time.perf_counter()
    h.predict_var(xq)
    print("predict_var points/s", 65536 / (time.perf_counter() - t0))
