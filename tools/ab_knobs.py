#!/usr/bin/env python3
"""A/B of run-time knobs inside ONE process (egx_set_tuning): the sweep's throughput (24 candidates in flight as three
lock-step groups of eight, 48 candidates per measurement) and one lock-step group's per-launch trailing-update rate
(lur_side = 0), for each setting of one knob, interleaved over several rounds.

    python tools/ab_knobs.py "potrf_left=0" "potrf_left=1" [--rounds 3] [--n 16384] [--d 32]
    python tools/ab_knobs.py "potrf_left=0,stream_min=128" "potrf_left=1,stream_min=8"     (several knobs per setting)"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egobox_amd as egx  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="+", help="knob=value[,knob=value...] per setting")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--d", type=int, default=32)
ap.add_argument("--in-flight", type=int, default=24)
ap.add_argument("--cands", type=int, default=48)
ap.add_argument("--no-group", action="store_true")
ap.add_argument("--lockstep", type=int, default=0)
a = ap.parse_args()
x, y = egx.workload.make_training_set(a.n, a.d, 42)
base = egx.workload.default_theta(a.d)
rng = np.random.default_rng(3)
h = egx.GpHandle(x, y, corr=0, n_workspaces=a.in_flight)
cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(a.cands, a.d))
if a.lockstep:
    h.set_lockstep(a.lockstep)
print(f"in flight {a.in_flight}, lock-step {h.set_lockstep(a.lockstep)}, {a.cands} candidates per measurement", flush=True)
h.likelihood_batch(cands[: a.in_flight])
a.values = a.settings


def apply(setting):
    for kv in setting.split(","):
        k, v = kv.split("=")
        egx.set_tuning(k.strip(), int(v))


res = {v: [] for v in a.values}
grp = {v: [] for v in a.values}
one = {v: [] for v in a.values}
for r in range(a.rounds):
    for v in a.values:
        apply(v)
        t0 = time.perf_counter()
        lk, st = h.likelihood_batch(cands * (1.0 + 1e-3 * r))
        res[v].append(a.cands / (time.perf_counter() - t0))
        assert np.all(st == 0), st
h.close()
h1 = egx.GpHandle(x, y, corr=0, n_workspaces=1)
h1.finalize(base)
for r in range(a.rounds):
    for v in a.values:
        apply(v)
        t0 = time.perf_counter()
        h1.finalize(base * (1.0 + 1e-3 * r))
        one[v].append((time.perf_counter() - t0) * 1e3)
h1.close()
if not a.no_group:
    g = egx.GpHandle(x, y, corr=0, n_workspaces=8)
    g.set_lockstep(8)
    egx.set_tuning("lur_side", 0)
    ths = np.stack([base * (1.0 + 0.01 * c) for c in range(8)])
    g.likelihood_batch(ths)
    for r in range(a.rounds):
        for v in a.values:
            apply(v)
            egx.set_tuning("lur_side", 0)
            g.likelihood_batch(ths * (1.0 + 1e-3 * r))
            t = g.timings()
            grp[v].append((t["syrk_flops"] / t["potrf_syrk_ms"] / 1e9, t["potrf_ms"]))
    g.close()
for v in a.values:
    line = (f"{v}: sweep fits/s {' '.join(f'{q:.2f}' for q in res[v])} (mean {np.mean(res[v]):.2f}); lone fit ms "
            f"{' '.join(f'{q:.2f}' for q in one[v])}")
    if grp[v]:
        line += (f"; group of 8 trailing-update launches TFLOP/s {' '.join(f'{q[0]:.2f}' for q in grp[v])}, "
                 f"group potrf ms {' '.join(f'{q[1]:.1f}' for q in grp[v])}")
    print(line, flush=True)
