#!/usr/bin/env python3
"""One slot of `ws` against two slots of ws / 2 in flight, by number of workspaces and size; `per` candidates per call."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx  # noqa: E402

for n, d in ((1024, 8), (2048, 8), (4096, 8), (8192, 16), (12288, 24)):
    x, y = egx.workload.make_training_set(n, d, 42)
    th = egx.workload.default_theta(d) * 3.0
    for ws in (4, 6, 8, 12, 16):
        if n >= 12288 and ws > 8:
            continue
        h = egx.GpHandle(x, y, corr=0, n_workspaces=ws)
        for per in (ws, 4 * ws):
            ths = np.stack([th * (1 + 0.01 * c) for c in range(per)])
            out = []
            for ls in (ws, (ws + 1) // 2, (ws + 2) // 3):
                got = h.set_lockstep(ls)
                h.likelihood_batch(ths)
                r = []
                for i in range(5):
                    t0 = time.perf_counter()
                    lk, st = h.likelihood_batch(ths * (1 + 1e-3 * i))
                    r.append(per / (time.perf_counter() - t0))
                out.append(f"width {got}: {np.median(r):8.1f}")
            print(f"n={n} workspaces {ws} candidates per call {per}: " + " | ".join(out) + " likelihoods/s", flush=True)
        h.close()
