#!/usr/bin/env python3
"""Where the time of egx_gp_create_group goes once the pool is warm (config 5's eight experts): create -> fit -> close, three times."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

k, n, d = 8, 8192, 16
sets = [egx.workload.make_training_set(n, d, 7 + e) for e in range(k)]
xs = np.stack([s[0] for s in sets])
ys = np.stack([s[1] for s in sets])
th = np.tile(egx.workload.default_theta(d), (k, 1))
for rep in range(4):
    t0 = time.perf_counter()
    hs = egx.GpHandle.create_group(xs, ys)
    t1 = time.perf_counter()
    egx.finalize_multi(hs, th)
    t2 = time.perf_counter()
    for h in hs:
        h.close()
    t3 = time.perf_counter()
    print(f"rep {rep}: create_group {1e3 * (t1 - t0):.2f} ms, finalize_multi {1e3 * (t2 - t1):.2f} ms, close {1e3 * (t3 - t2):.2f} ms", flush=True)
# one lone handle the same way
x, y = sets[0]
for rep in range(3):
    t0 = time.perf_counter()
    h = egx.GpHandle(x, y, corr=0, n_workspaces=1)
    t1 = time.perf_counter()
    h.finalize(th[0])
    t2 = time.perf_counter()
    h.close()
    t3 = time.perf_counter()
    print(f"lone rep {rep}: create {1e3 * (t1 - t0):.2f} ms, finalize {1e3 * (t2 - t1):.2f} ms, close {1e3 * (t3 - t2):.2f} ms", flush=True)
