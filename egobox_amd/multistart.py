"""Multistart candidates for the theta optimisation (crates/gp/src/optimization.rs:26-71).

`(n_start + 1) x h` matrix in log10 space: row 0 = log10 of the user theta0, rows 1.. = an LHS in the
log10 bounds.  The reference draws that LHS with `Lhs(kind=Maximin, rng=Xoshiro256Plus::seed_from_u64(42))`
(egobox-doe, outside the accelerated path); bit parity with that stream is not required, so a
seeded maximin-by-restarts classic LHS stands in.  These rows are also the definition of the theta-sweep
inputs (BASELINE config 4).
"""
from __future__ import annotations

import numpy as np


def lhs_classic(n, d, rng):
    """Classic LHS on [0,1]^d: one point per stratum and per column (crates/doe/src/lhs.rs:236-257)."""
    x = np.empty((n, d))
    for j in range(d):
        x[:, j] = (rng.permutation(n) + rng.random(n)) / n
    return x


def lhs_maximin(n, d, rng, tries=5):
    """Best of `tries` classic LHS by minimum pairwise distance (crates/doe/src/lhs.rs:276-297 idea)."""
    best, best_d = None, -1.0
    for _ in range(max(1, tries)):
        x = lhs_classic(n, d, rng)
        if 1 < n <= 2048:
            diff = x[:, None, :] - x[None, :, :]
            dist = np.sqrt((diff ** 2).sum(-1)) + np.eye(n) * 1e9
            dm = dist.min()
        else:
            dm = 0.0
        if dm > best_d:
            best, best_d = x, dm
    return best


def prepare_multistart(n_start, theta0, bounds, seed=42):
    """optimization.rs:26-71.  Returns (theta0s_log10 ((n_start+1) x h), bounds_log10)."""
    theta0 = np.atleast_1d(np.asarray(theta0, dtype=np.float64))
    h = theta0.size
    bl = [(np.log10(lo), np.log10(hi)) for lo, hi in bounds]
    if len(bl) != h:
        raise ValueError(f"bounds length {len(bl)} != theta length {h}")
    out = np.zeros((n_start + 1, h))
    out[0] = np.log10(theta0)
    if n_start == 1:
        rng = np.random.default_rng()  # the reference uses an entropy-seeded draw here (optimization.rs:44)
        out[1] = [rng.uniform(a, b) for a, b in bl]
    elif n_start > 1:
        rng = np.random.default_rng(seed)
        u = lhs_maximin(n_start, h, rng)
        lo = np.array([a for a, _ in bl])
        hi = np.array([b for _, b in bl])
        out[1:] = lo + u * (hi - lo)
    return out, bl


def theta_sweep_candidates(k, h, theta0=0.1, bounds=(1e-2, 1e1), seed=42):
    """BASELINE config 4: row 0 = theta0 on every dimension, k-1 rows log-uniform LHS in the bounds.
    Returned in LINEAR theta units, shape (k, h)."""
    starts, _ = prepare_multistart(k - 1, np.full(h, theta0), [bounds] * h, seed=seed)
    return 10.0 ** starts
