#!/usr/bin/env python3
"""Diagnostic (round 6): does a tuned fit land on the same optimum whatever the handle's workspaces (schedule row)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import egobox_amd as egx

def _data(n, d, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(size=(n, d))
    y = np.sin(3 * x[:, 0]) + x[:, 1:].sum(axis=1) ** 2 + 0.1 * rng.standard_normal(n)
    return x, y

n, d = 2100, 4
for seed in (90, 91):
    x, y = _data(n, d, seed)
    for me in (30, 100, 300):
        for nws in (1, 2, 4):
            p = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()).n_start(3).max_eval(me).n_workspaces(nws)
            g = p.fit(x, y)
            print(f"seed {seed} max_eval {me} workspaces {nws}: lkh {g.likelihood():.10f} evals {g.n_evals} theta {g._h.inner()['theta']} schedule {g._h.schedule()}", flush=True)
            g.close()
