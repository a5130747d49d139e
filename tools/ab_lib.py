#!/usr/bin/env python3
"""A/B of LIBRARY BUILDS (tools/dev_build.sh <name> <defines> -> egobox_amd/lib/_dev/libegx_gp_hip_<name>.so), one child process
per build and round, interleaved: the headline sweep (n = 16384, lock-step 8, 16 in flight), one lock-step group alone (the
launches of bench.py's `roofline`), lone fits at n = 16384 / 8192 / 4096, predict_var at config 5's expert.
    python tools/ab_lib.py product ilv0 [--rounds 2]        ("product" = egobox_amd/lib/libegx_gp_hip.so)"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    import numpy as np
    sys.path.insert(0, ROOT)
    import egobox_amd as egx
    out = {}
    n, d = 16384, 32
    x, y = egx.workload.make_training_set(n, d, 42)
    base = egx.workload.default_theta(d)
    rng = np.random.default_rng(3)
    h = egx.GpHandle(x, y, corr=0, n_workspaces=16)
    h.set_lockstep(8)
    cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(48, d))
    h.likelihood_batch(cands[:16])
    t0 = time.perf_counter()
    lk, st = h.likelihood_batch(cands)
    out["sweep_fits_per_s"] = 48 / (time.perf_counter() - t0)
    out["sweep_checksum"] = float(lk.sum())
    h.close()
    egx.trim()
    g = egx.GpHandle(x, y, corr=0, n_workspaces=8)
    g.set_lockstep(8)
    egx.set_tuning("lur_side", 0)
    ths = np.stack([base * (1.0 + 0.01 * c) for c in range(8)])
    g.likelihood_batch(ths)
    g.likelihood_batch(ths * 1.001)
    t = g.timings()
    out["group_long_update_tflops"] = t["syrk_flops"] / t["potrf_syrk_ms"] / 1e9
    out["group_potrf_ms"] = t["potrf_ms"]
    g.close()
    egx.set_tuning("lur_side", 1)
    egx.trim()
    for nn, dd in ((16384, 32), (8192, 16), (4096, 8)):
        xx, yy = egx.workload.make_training_set(nn, dd, 42)
        th = egx.workload.default_theta(dd)
        h1 = egx.GpHandle(xx, yy, corr=0, n_workspaces=1)
        h1.finalize(th)
        ts = []
        for r in range(5):
            t0 = time.perf_counter()
            h1.finalize(th * (1.0 + 1e-3 * r))
            ts.append((time.perf_counter() - t0) * 1e3)
        out[f"lone_fit_ms_n{nn}"] = float(np.median(ts))
        out[f"lone_potrf_ms_n{nn}"] = h1.timings()["potrf_ms"]
        if nn == 8192:
            xq = np.random.default_rng(1).uniform(size=(32768, dd))
            h1.predict_var(xq[:1024])
            t0 = time.perf_counter()
            h1.predict_var(xq)
            out["predict_var_points_per_s_n8192"] = 32768 / (time.perf_counter() - t0)
        h1.close()
        egx.trim()
    print("AB " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("builds", nargs="+")
    ap.add_argument("--rounds", type=int, default=2)
    a = ap.parse_args()
    res = {b: [] for b in a.builds}
    for r in range(a.rounds):
        for b in a.builds:
            env = dict(os.environ)
            env.pop("EGX_TEST_LIBRARY", None)
            if b != "product":
                env["EGX_TEST_LIBRARY"] = b
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("AB ")]
            if not line:
                print(f"{b} round {r}: failed\n{p.stdout[-500:]}\n{p.stderr[-800:]}", flush=True)
                continue
            res[b].append(json.loads(line[0][3:]))
    keys = list(next(v for v in res.values() if v)[0].keys())
    for k in keys:
        print(k + ": " + " | ".join(f"{b}: " + " ".join(f"{q[k]:.6g}" for q in res[b]) for b in a.builds), flush=True)
