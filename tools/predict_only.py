#!/usr/bin/env python3
"""One expert of BASELINE config 5 (n = 8192, d = 16): fit, then predict and predict_var on 100 000 points -- the command the
predict-side PMC pass profiles (tools/gpu_pmc.sh predict)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

x, y = egx.workload.make_training_set(8192, 16, 7)
xq = np.random.default_rng(7).random((100000, 16))
with egx.GpHandle(x, y) as h:
    h.finalize(egx.workload.default_theta(16))
    for _ in range(3):
        h.predict(xq)
    h.predict_var(xq[:32768])
