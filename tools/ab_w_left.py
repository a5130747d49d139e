#!/usr/bin/env python3
"""A/B of the C^-T rider's left-looking update (EGX_W_LEFT, kernels_chol.hip launch_potrf): likelihood + theta-gradient of
one candidate and of a lock-step batch, both settings in ONE process; the gradients of the settings are compared bit by bit.
    tools/ab_w_left.py n d corr n_workspaces lockstep setting [setting ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

n, d, corr, nws, width = (int(a) for a in sys.argv[1:6])
settings = [int(a) for a in sys.argv[6:]] or [1, 0]
x, y = egx.workload.make_training_set(n, d, 42)
th = egx.workload.default_theta(d)
flop = float(n) ** 3
ths = np.stack([th * (1 + 0.003 * c) for c in range(max(nws, 1))])
res = {}
with egx.GpHandle(x, y, corr=corr, n_workspaces=nws) as h:
    print(f"n={n} d={d} corr={corr} workspaces={nws} lock-step width {h.set_lockstep(width)}", flush=True)
    for rnd in range(2):
        for wl in settings:
            prev = egx.set_tuning("w_left", wl)
            try:
                if rnd == 0:
                    h.likelihood_grad_batch(ths)  # scratch / warm-up
                t0 = time.perf_counter()
                lks, gs, sts = h.likelihood_grad_batch(ths)
                dt = time.perf_counter() - t0
                t1 = time.perf_counter()
                lk1, g1, st1 = h.likelihood_grad(ths[-1])
                dt1 = time.perf_counter() - t1
            finally:
                egx.set_tuning("w_left", prev)
            res[wl] = (lks, gs)
            print(f"  w_left={wl}: batch of {len(ths)} {dt / len(ths) * 1e3:.3f} ms per candidate = {len(ths) * flop / dt / 78.6e12:.3f} "
                  f"of peak, ok {int(np.sum(sts == 0))}; one candidate {dt1 * 1e3:.3f} ms = {flop / dt1 / 78.6e12:.3f}; single == batch "
                  f"bits {bool(lk1 == lks[-1] and np.array_equal(g1, gs[-1]))}", flush=True)
    a, b = res[settings[0]], res[settings[-1]]
    print("  same bits under both settings:", bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])), flush=True)
