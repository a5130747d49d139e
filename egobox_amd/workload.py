"""Synthetic workload of SURVEY.md section 8(d): seeded classic LHS inputs and the Griewank function the
reference's own benches use (crates/gp/benches/gp.rs:21-25, python/egobox/tests/test_gpmix.py:10-21)."""
from __future__ import annotations

import numpy as np

from .multistart import lhs_classic


def lhs(n, d, seed):
    return lhs_classic(n, d, np.random.default_rng(seed))


def griewank(x01):
    """Griewank on [-600, 600]^d, inputs given on the unit cube."""
    x = -600.0 + 1200.0 * np.asarray(x01, dtype=np.float64)
    i = np.arange(1, x.shape[1] + 1)
    return (x * x).sum(axis=1) / 4000.0 - np.prod(np.cos(x / np.sqrt(i)), axis=1) + 1.0


def make_training_set(n, d, seed=42):
    x = lhs(n, d, seed)
    return x, griewank(x)


def default_theta(d):
    """theta_j = 0.5/sqrt(d): typical off-diagonal correlation ~ exp(-1/4) on unit-variance inputs."""
    return np.full(d, 0.5 / np.sqrt(d))
