#!/usr/bin/env python3
"""Fit time by problem size, one and two fits in flight (env-switchable variants are A/B'd by the caller):
    EGX_POTRF_GROUP=4 python tools/size_ab.py 4096 8192 16384"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

pool = ThreadPoolExecutor(2)
for n in [int(a) for a in sys.argv[1:]] or [4096]:
    d = 8 if n <= 4096 else (16 if n <= 8192 else 32)
    x, y = egx.workload.make_training_set(n, d, 42)
    th = egx.workload.default_theta(d) * (4.0 if n <= 4096 else 1.0)
    hs = [egx.GpHandle(x, y) for _ in range(2)]
    reps = 20 if n <= 8192 else 6
    for h in hs:
        h.finalize(th)
    t0 = time.perf_counter()
    for _ in range(reps):
        hs[0].finalize(th)
    t1 = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        list(pool.map(lambda h: h.finalize(th), hs))
    t2 = (time.perf_counter() - t0) / reps / 2
    print(f"n={n} d={d}: one in flight {t1 * 1e3:.3f} ms/fit, two in flight {t2 * 1e3:.3f} ms/fit, potrf {hs[0].timings()['potrf_ms']:.3f} ms",
          flush=True)
    for h in hs:
        h.close()
