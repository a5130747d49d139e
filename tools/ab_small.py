#!/usr/bin/env python3
"""A/B of run-time knobs at a SMALL size (config 2: n = 4096, d = 8): one fit in flight (latency: the serial chain) and a
lock-step batch of twelve (a tuned fit's round), per knob setting, interleaved.
    python tools/ab_small.py "pipe=0" "pipe=1" [--n 4096] [--d 8]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+")
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--d", type=int, default=8)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()


def apply(setting):
    for kv in setting.split(","):
        k, v = kv.split("=")
        egx.set_tuning(k.strip(), int(v))


x, y = egx.workload.make_training_set(a.n, a.d, 42)
th = egx.workload.default_theta(a.d) * 3.0
h1 = egx.GpHandle(x, y, corr=0, n_workspaces=1)
hb = egx.GpHandle(x, y, corr=0, n_workspaces=12)
ths = np.stack([th * (1 + 0.01 * c) for c in range(48)])
h1.likelihood(th)
hb.likelihood_batch(ths)
lone = {s: [] for s in a.settings}
potrf = {s: [] for s in a.settings}
batch = {s: [] for s in a.settings}
for r in range(a.rounds):
    for s in a.settings:
        apply(s)
        t0 = time.perf_counter()
        for j in range(20):
            h1.likelihood(th * (1 + 1e-3 * j))
        lone[s].append((time.perf_counter() - t0) / 20 * 1e3)
        potrf[s].append(h1.timings()["potrf_ms"])
        t0 = time.perf_counter()
        lk, st = hb.likelihood_batch(ths * (1 + 1e-3 * r))
        batch[s].append(len(ths) / (time.perf_counter() - t0))
        assert np.all(st == 0)
for s in a.settings:
    print(f"{s}: one evaluation in flight {' '.join(f'{v:.3f}' for v in lone[s])} ms (potrf {' '.join(f'{v:.3f}' for v in potrf[s])} ms = "
          f"{a.n ** 3 / 3 / np.mean(potrf[s]) / 1e9:.1f} TFLOP/s); lock-step 12: {' '.join(f'{v:.0f}' for v in batch[s])} fits/s", flush=True)
h1.close()
hb.close()
