#!/usr/bin/env python3
"""First handle of a shape (pool miss): creation / first fit / second fit, by size and workspaces -- what a mixture whose clusters
differ in size pays per expert (crates/moe/src/algorithm.rs:167-177)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

x0, y0 = egx.workload.make_training_set(1000, 8, 1)
with egx.GpHandle(x0, y0, corr=1) as h:
    h.finalize(np.full(8, 1.0))  # code objects, runtime pools
for nws in (1, 12):
    for n in (2000, 2300, 4096, 4500, 8192, 8500, 16384):
        if nws > 1 and n > 8500:
            continue
        d = 8
        x, y = egx.workload.make_training_set(n, d, n)
        th = np.full(d, 1.0)
        t0 = time.perf_counter()
        h = egx.GpHandle(x, y, corr=1, n_workspaces=nws)
        t1 = time.perf_counter()
        h.finalize(th)
        t2 = time.perf_counter()
        h.finalize(th)
        t3 = time.perf_counter()
        h.close()
        print(f"n={n} workspaces {nws}: first create {1e3 * (t1 - t0):7.2f} ms, first fit {1e3 * (t2 - t1):7.2f} ms, second fit {1e3 * (t3 - t2):7.2f} ms", flush=True)
egx.trim()
