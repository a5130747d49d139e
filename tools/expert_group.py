#!/usr/bin/env python3
"""Eight experts of config 5's shape (n = 8192, d = 16) fitted in lock-step: egx_gp_create_group + egx_gp_finalize_multi a few
times, nothing else -- the command profiled for profiles/r05_expert_group_*.txt
(`rocprofv3 --kernel-trace --stats -- python tools/expert_group.py [k] [n] [d] [reps]`)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
d = int(sys.argv[3]) if len(sys.argv) > 3 else 16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
sets = [egx.workload.make_training_set(n, d, 7 + e) for e in range(k)]
xs = np.stack([s[0] for s in sets])
ys = np.stack([s[1] for s in sets])
th = np.tile(egx.workload.default_theta(d), (k, 1))
t0 = time.perf_counter()
hs = egx.GpHandle.create_group(xs, ys)
print(f"create_group: {1e3 * (time.perf_counter() - t0):.2f} ms", flush=True)
for i in range(reps):
    t0 = time.perf_counter()
    egx.finalize_multi(hs, th * (1 + 0.01 * i))
    print(f"finalize_multi {i}: {1e3 * (time.perf_counter() - t0):.3f} ms for {k} models", flush=True)
for i in range(2):
    t0 = time.perf_counter()
    lk, st = egx.likelihood_multi(hs, th * (1.1 + 0.01 * i))
    print(f"likelihood_multi {i}: {1e3 * (time.perf_counter() - t0):.3f} ms, statuses ok {int((st == 0).sum())}", flush=True)
for h in hs:
    h.close()
