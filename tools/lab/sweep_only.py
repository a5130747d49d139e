#!/usr/bin/env python3
"""the headline sweep only (n = 16384, lock-step 8, 16 in flight, 48 candidates x 3) and a lone fit: stream / queue mapping A/Bs"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx
n, d = 16384, 32
x, y = egx.workload.make_training_set(n, d, 42)
base = egx.workload.default_theta(d)
rng = np.random.default_rng(3)
h = egx.GpHandle(x, y, corr=0, n_workspaces=16)
h.set_lockstep(8)
cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(48, d))
h.likelihood_batch(cands[:16])
r = []
for i in range(3):
    t0 = time.perf_counter(); lk, st = h.likelihood_batch(cands); r.append(48 / (time.perf_counter() - t0))
h.close(); egx.trim()
h1 = egx.GpHandle(x, y, corr=0, n_workspaces=1)
h1.finalize(base); ts = []
for i in range(4):
    t0 = time.perf_counter(); h1.finalize(base * (1 + 1e-3 * i)); ts.append(1e3 * (time.perf_counter() - t0))
print("sweep fits/s " + " ".join(f"{q:.2f}" for q in r) + f" | lone fit ms {np.median(ts):.2f} potrf {h1.timings()['potrf_ms']:.2f}", flush=True)
