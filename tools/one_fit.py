#!/usr/bin/env python3
"""A few fixed-theta fits at one size, nothing else: the command profiled for timelines
(`rocprofv3 --kernel-trace -- python tools/one_fit.py [n] [d] [reps] [corr] [width] [workspaces]`; width > 1: likelihood batches of that many
candidates on a handle with as many workspaces, in lock-step -- a round of a tuned fit's starts)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
corr = int(sys.argv[4]) if len(sys.argv) > 4 else 0  # 0 sq-exp, 1 abs-exp, 2 Matern-3/2, 3 Matern-5/2
x, y = egx.workload.make_training_set(n, d, 42)
th = egx.workload.default_theta(d)
width = int(sys.argv[5]) if len(sys.argv) > 5 else 1
nws = int(sys.argv[6]) if len(sys.argv) > 6 else width  # workspaces (candidates per batch): nws / width slots in flight
if width > 1:
    h = egx.GpHandle(x, y, corr=corr, n_workspaces=nws)
    h.set_lockstep(width)
    print("schedule", h.schedule(), flush=True)
    for i in range(reps):
        t0 = time.perf_counter()
        lk, st = h.likelihood_batch(np.stack([th * (1 + 0.01 * (i * nws + c)) for c in range(nws)]))
        print(f"batch {i}: {1e3 * (time.perf_counter() - t0):.3f} ms for {nws} candidates in slots of {width}, statuses ok {int((st == 0).sum())}  {h.timings()}",
              flush=True)
    h.close()
    sys.exit(0)
h = egx.GpHandle(x, y, corr=corr)
print("schedule", h.schedule(), flush=True)
for i in range(reps):
    t0 = time.perf_counter()
    h.finalize(th * (1 + 0.01 * i))
    print(f"fit {i}: {1e3 * (time.perf_counter() - t0):.3f} ms  {h.timings()}", flush=True)
h.close()
