#!/usr/bin/env python3
"""likelihood + theta-gradient at one size: single candidate and lock-step batches (config 3: n = 16384, d = 32, Matern-5/2).
`rocprofv3 --kernel-trace --stats -- python tools/grad_bench.py 16384 32 3 1` profiles the single-candidate path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 32
corr = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nws = int(sys.argv[4]) if len(sys.argv) > 4 else 8
x, y = egx.workload.make_training_set(n, d, 42)
th = egx.workload.default_theta(d)
flop = float(n) ** 3
h = egx.GpHandle(x, y, corr=corr, n_workspaces=nws)
h.likelihood_grad(th * 0.99)
for j in range(3):
    t0 = time.perf_counter()
    lk, g, st = h.likelihood_grad(th * (1 + 0.004 * j))
    dt = time.perf_counter() - t0
    print(f"single {j}: {dt * 1e3:.2f} ms = {flop / dt / 1e12:.1f} TFLOP/s ({flop / dt / 78.6e12:.3f} of peak), lk {lk:.6f} |g| {np.linalg.norm(g):.6e} st {st}",
          flush=True)
for k, w in ((nws, nws), (2 * nws, nws), (nws, max(1, nws // 2))):
    if nws < 2:
        break
    h.set_lockstep(w)
    ths = np.stack([th * (1 + 0.003 * c) for c in range(k)])
    if k == nws and w == nws:
        h.likelihood_grad_batch(ths)  # scratch for all workspaces
    t0 = time.perf_counter()
    lks, gs, sts = h.likelihood_grad_batch(ths)
    dt = time.perf_counter() - t0
    print(f"batch of {k}, lock-step {w}: {dt / k * 1e3:.2f} ms per candidate = {k * flop / dt / 1e12:.1f} TFLOP/s "
          f"({k * flop / dt / 78.6e12:.3f} of peak), ok {int(np.sum(sts == 0))}", flush=True)
t0 = time.perf_counter()
h.finalize(th)
print(f"fit alone (finalize): {(time.perf_counter() - t0) * 1e3:.2f} ms", flush=True)
h.close()
