#!/usr/bin/env python3
"""First group of a shape (pool miss) by number of members, and one lone handle per such shape for comparison."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

x0, y0 = egx.workload.make_training_set(1000, 8, 1)
with egx.GpHandle(x0, y0, corr=1) as h:
    h.finalize(np.full(8, 1.0))
d = 8
for k, n in ((1, 5000), (2, 5300), (4, 5600), (8, 5900), (8, 8192), (1, 6200), (8, 2100)):
    sets = [egx.workload.make_training_set(n, d, 7 + e) for e in range(k)]
    xs = np.stack([s[0] for s in sets])
    ys = np.stack([s[1] for s in sets])
    t0 = time.perf_counter()
    hs = egx.GpHandle.create_group(xs, ys, corr=1)
    t1 = time.perf_counter()
    egx.finalize_multi(hs, np.tile(np.full(d, 1.0), (k, 1)))
    t2 = time.perf_counter()
    for h in hs:
        h.close()
    t3 = time.perf_counter()
    print(f"k={k} n={n}: first create_group {1e3 * (t1 - t0):7.2f} ms ({1e3 * (t1 - t0) / k:6.2f} per member), first fit {1e3 * (t2 - t1):6.2f} ms, close {1e3 * (t3 - t2):5.2f} ms", flush=True)
