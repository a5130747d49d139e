#!/usr/bin/env python3
"""A few likelihood + theta-gradient evaluations of ONE candidate at one size, nothing else: the command profiled for the
gradient's kernel stats / PMC passes (`rocprofv3 --kernel-trace --stats -- python tools/one_grad.py [n] [d] [reps] [corr]`)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
corr = int(sys.argv[4]) if len(sys.argv) > 4 else 3
x, y = egx.workload.make_training_set(n, d, 42)
th = egx.workload.default_theta(d)
h = egx.GpHandle(x, y, corr=corr, n_workspaces=1)
for i in range(reps):
    t0 = time.perf_counter()
    lk, g, st = h.likelihood_grad(th * (1 + 0.004 * i))
    print(f"grad {i}: {1e3 * (time.perf_counter() - t0):.3f} ms  lk {lk:.6f} |g| {np.linalg.norm(g):.6e} st {st}", flush=True)
h.close()
