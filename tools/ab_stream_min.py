#!/usr/bin/env python3
"""A/B of the tile count from which an update launch takes the stream kernel ("stream_min", read at launch time) where the
launches carry SEVERAL matrices (grid.z): the headline sweep (n = 16384, lock-step groups of eight, sixteen in flight), the
eight experts of config 5 as a group (8 x n = 8192) and a lock-step batch of twelve at n = 8192 -- interleaved, one process.
    python tools/ab_stream_min.py 128 32 16 8 [--rounds 3]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("values", nargs="+", type=int)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--skip", default="")
a = ap.parse_args()
res = {}


def note(leg, v, val):
    res.setdefault(leg, {}).setdefault(v, []).append(val)


if "sweep" not in a.skip:
    n, d = 16384, 32
    x, y = egx.workload.make_training_set(n, d, 42)
    base = egx.workload.default_theta(d)
    rng = np.random.default_rng(3)
    h = egx.GpHandle(x, y, corr=0, n_workspaces=16)
    h.set_lockstep(8)
    cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(48, d))
    h.likelihood_batch(cands[:16])
    ref = None
    for r in range(a.rounds):
        for v in a.values:
            egx.set_tuning("stream_min", v)
            t0 = time.perf_counter()
            lk, st = h.likelihood_batch(cands)
            note("sweep n=16384 lock-step 8, 16 in flight: fits/s", v, 48 / (time.perf_counter() - t0))
            assert np.all(st == 0), st
            if ref is None:
                ref = lk
            note("  max rel diff of the likelihoods to the first setting", v, float(np.max(np.abs(lk - ref) / np.abs(ref))))
    h.close()
    egx.trim()

if "experts" not in a.skip:
    k, n, d = 8, 8192, 16
    sets = [egx.workload.make_training_set(n, d, 7 + e) for e in range(k)]
    xs = np.stack([s[0] for s in sets])
    ys = np.stack([s[1] for s in sets])
    th = np.tile(egx.workload.default_theta(d), (k, 1))
    hs = egx.GpHandle.create_group(xs, ys)
    egx.finalize_multi(hs, th)
    for r in range(a.rounds):
        for v in a.values:
            egx.set_tuning("stream_min", v)
            t0 = time.perf_counter()
            egx.finalize_multi(hs, th * (1 + 0.01 * r))
            note("8 experts n=8192 as a group, refit: ms", v, 1e3 * (time.perf_counter() - t0))
    for h in hs:
        h.close()
    egx.trim()

if "batch12" not in a.skip:
    for n, d in ((8192, 16), (4096, 8)):
        x, y = egx.workload.make_training_set(n, d, 42)
        base = egx.workload.default_theta(d)
        rng = np.random.default_rng(5)
        h = egx.GpHandle(x, y, corr=0, n_workspaces=12)
        h.set_lockstep(12)
        cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(24, d))
        h.likelihood_batch(cands)
        for r in range(a.rounds):
            for v in a.values:
                egx.set_tuning("stream_min", v)
                t0 = time.perf_counter()
                lk, st = h.likelihood_batch(cands)
                note(f"n={n} lock-step 12, 24 candidates: likelihoods/s", v, 24 / (time.perf_counter() - t0))
        h.close()
        egx.trim()

for leg, by in res.items():
    for v, xs_ in by.items():
        print(f"{leg} | stream_min={v}: {' '.join(f'{q:.4g}' for q in xs_)} (mean {np.mean(xs_):.4g})", flush=True)
