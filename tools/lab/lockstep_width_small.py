#!/usr/bin/env python3
"""A/B at small sizes: the same batch of candidates on one handle with `ws` workspaces, by lock-step width (slots of that
width are in flight side by side on their own streams).  likelihoods/s, median of `rounds`.
    python tools/lab/lockstep_width_small.py 4096 8 48"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 48
rounds = 5
x, y = egx.workload.make_training_set(n, d, 42)
th = egx.workload.default_theta(d) * 3.0
ths = np.stack([th * (1 + 0.01 * c) for c in range(nc)])
for ws in (12, 24):
    h = egx.GpHandle(x, y, corr=0, n_workspaces=ws)
    for ls in (12, 8, 6, 4, 3, 2):
        if ls > ws:
            continue
        got = h.set_lockstep(ls)
        h.likelihood_batch(ths[:ws])
        r = []
        for i in range(rounds):
            t0 = time.perf_counter()
            lk, st = h.likelihood_batch(ths * (1 + 1e-3 * i))
            r.append(nc / (time.perf_counter() - t0))
            assert np.all(st == 0)
        print(f"n={n} d={d} workspaces {ws} lock-step {got} ({(ws + got - 1) // got} slots in flight): "
              f"{np.median(r):8.1f} likelihoods/s  ({' '.join(f'{v:.0f}' for v in r)})  schedule {h.schedule()}", flush=True)
    h.close()
