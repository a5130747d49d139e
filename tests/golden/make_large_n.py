#!/usr/bin/env python3
"""Oracle fixtures at the BASELINE.json configuration sizes (run in the BUILD container, CPU only).

The numpy/scipy oracle (`oracle/gp_oracle.py`, pinned to the reference's golden vectors by
tests/test_oracle_golden.py) needs ~100 s per likelihood at n = 16384 on 8 cores, too slow for the
`-m gpu` test budget, so its outputs at the headline sizes are committed as DATA:

  fit_n16384_d32_<corr>   config 3 / metric line: likelihood, sigma2, beta + 1000 predictions and variances
                          (sq-exp and Matern-5/2, seed 42, theta = 0.5/sqrt(d))
  grad_n4096_d32_<corr>   theta-gradient of the likelihood (oracle closed form), the largest n whose
                          (pairs, d) table fits comfortably
  grad_n16384_d32_matern52  the same at config 3's full size (--only grad16384: one n^3 CPU job, ~10 minutes)
  sweep_n16384_d32        config 4: likelihood + status for 28 rows of theta_sweep_candidates(512, 32)
                          (row 0, the first 13 LHS rows, the 14 rows with the smallest sum theta^2) and 3
                          extra lower-bound rows that exercise the not-positive-definite status
  expert_n8192_d16        config 5: one expert (seed 7) with 1000 predict / predict_var points
  experts_1_to_7_n8192_d16  config 5: the other seven experts (seeds 8..14) on the first 100 query points, and the
                          smooth / hard mixture of all eight oracle experts on those points (--only experts)

Inputs are regenerated in the tests from the same seeds (oracle.lhs_classic / griewank ==
egobox_amd.workload), so only thetas, scalars and the prediction vectors are stored.

  xgrad_n200_d300_<corr>  x-gradients of mean and variance (oracle's restated jacobians, algorithm.rs:510-617) at d = 300 --
                          beyond the 256 dimensions a training slab of 64 points fits LDS with -- on query rows 0 and 1 of
                          tests/test_gpu_parity.py::test_x_gradients_beyond_256_dimensions (--only xgrad300; the oracle's
                          product-over-all-but-one Matern jacobian costs minutes per point at this d)

    python tests/golden/make_large_n.py [--only fit|grad|grad16384|sweep|expert|experts|xgrad300] [--out tests/golden/large_n.json]

Parts are merged into an existing output file, so the script can be run piecewise.
"""
import argparse
import importlib.util
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gp_oracle as O  # noqa: E402


def _multistart():
    # the pure-numpy candidate generator of the product, loaded WITHOUT importing the package (no .so needed)
    spec = importlib.util.spec_from_file_location("egx_multistart", os.path.join(ROOT, "egobox_amd", "multistart.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def queries(m, d, seed):
    return np.random.default_rng(seed).random((m, d))


def part_fit(out):
    n, d = 16384, 32
    x = O.lhs_classic(n, d, 42)
    y = O.griewank(x)
    theta = np.full(d, 0.5 / math.sqrt(d))
    xq = queries(1000, d, 7)
    # 200 of the queries sit on / next to training points (variance -> 0, the clamp region)
    xq[:100] = x[:100]
    xq[100:200] = x[100:200] + 1e-4
    for corr in (O.SQEXP, O.MATERN52):
        if f"fit_n{n}_d{d}_{corr}" in out:
            continue
        t0 = time.time()
        gp = O.fit_fixed(x, y, theta, O.CONSTANT, corr)
        diag = np.diag(gp.inner.r_chol)
        rec = {
            "n": n, "d": d, "seed": 42, "corr": corr, "mean": O.CONSTANT, "theta": theta.tolist(),
            "likelihood": gp.likelihood, "sigma2": gp.inner.sigma2, "beta": gp.inner.beta[:, 0].tolist(),
            "min_pivot": float(diag.min()), "gamma_head": gp.inner.gamma[:8, 0].tolist(),
            "gamma_norm": float(np.linalg.norm(gp.inner.gamma)),
            "query_seed": 7, "query_note": "rows 0..99 = x[0..99], rows 100..199 = x[100..199] + 1e-4, rest uniform",
            "predict": gp.predict(xq).tolist(), "predict_var": gp.predict_var(xq).tolist(),
        }
        out[f"fit_n{n}_d{d}_{corr}"] = rec
        print(f"fit {corr}: lkh {gp.likelihood!r} min pivot {diag.min():.3e} ({time.time() - t0:.0f}s)", flush=True)
        del gp
        yield  # checkpoint


def part_grad(out):
    n, d = 4096, 32
    x = O.lhs_classic(n, d, 42)
    y = O.griewank(x)
    theta = np.full(d, 0.5 / math.sqrt(d)) * (1.0 + 0.3 * np.sin(np.arange(d)))
    for corr in (O.SQEXP, O.MATERN52):
        t0 = time.time()
        lk, g = O.likelihood_grad(x, y, theta, O.CONSTANT, corr)
        out[f"grad_n{n}_d{d}_{corr}"] = {"n": n, "d": d, "seed": 42, "corr": corr, "theta": theta.tolist(),
                                          "likelihood": lk, "grad": g.tolist()}
        print(f"grad {corr}: lkh {lk!r} |g| {np.linalg.norm(g):.6e} ({time.time() - t0:.0f}s)", flush=True)


def part_grad16384(out):
    """Config 3 at its full size: the oracle's closed-form theta-gradient at n = 16384, d = 32, Matern-5/2 (one n^3 CPU job:
    ~2 GB each for R, C^-1, R^-1 and the per-dimension temporaries; ~10 minutes on 8 cores)."""
    n, d = 16384, 32
    x = O.lhs_classic(n, d, 42)
    y = O.griewank(x)
    theta = np.full(d, 0.5 / math.sqrt(d)) * (1.0 + 0.3 * np.sin(np.arange(d)))
    t0 = time.time()
    lk, g = O.likelihood_grad(x, y, theta, O.CONSTANT, O.MATERN52)
    out[f"grad_n{n}_d{d}_{O.MATERN52}"] = {"n": n, "d": d, "seed": 42, "corr": O.MATERN52, "theta": theta.tolist(),
                                             "likelihood": lk, "grad": g.tolist()}
    print(f"grad16384 {O.MATERN52}: lkh {lk!r} |g| {np.linalg.norm(g):.6e} ({time.time() - t0:.0f}s)", flush=True)


def part_sweep(out):
    n, d = 16384, 32
    x = O.lhs_classic(n, d, 42)
    y = O.griewank(x)
    cands = _multistart().theta_sweep_candidates(512, d)
    order = np.argsort((cands ** 2).sum(axis=1))
    rows = list(range(14))
    rows += [int(r) for r in order if int(r) not in rows][:14]
    extra = [np.full(d, v) for v in (0.01, 0.02, 0.03, 1e-3, 1e-4)]
    key = f"sweep_n{n}_d{d}"
    rec = out.get(key) or {"n": n, "d": d, "seed": 42, "corr": O.SQEXP, "candidates": "theta_sweep_candidates(512, 32)",
                           "rows": rows, "thetas": [cands[r].tolist() for r in rows] + [e.tolist() for e in extra],
                           "extra_rows_note": "last 5 thetas = all 0.01 / 0.02 / 0.03 (lower bound corner: cond(R) ~ 1 / nugget, "
                                              "ill posed) and all 1e-3 / 1e-4 (below the bounds: R ~ all ones, not positive "
                                              "definite), NOT in the 512-row list: the status channel at size",
                           "likelihood": [], "status": []}
    want = [cands[r].tolist() for r in rows] + [e.tolist() for e in extra]
    if len(rec["thetas"]) < len(want):  # extend a fixture written with fewer extra rows
        rec["thetas"] = rec["thetas"] + want[len(rec["thetas"]):]
        rec["extra_rows_note"] = ("last 5 thetas = all 0.01 / 0.02 / 0.03 (lower bound corner: cond(R) ~ 1 / nugget, ill "
                                  "posed) and all 1e-3 / 1e-4 (below the bounds: R ~ all ones, not positive definite), NOT "
                                  "in the 512-row list: the status channel at size")
    rec["n_extra"] = len(extra)
    thetas = np.array(rec["thetas"])
    for i in range(len(rec["likelihood"]), len(thetas)):
        t0 = time.time()
        lk, st = O.likelihood_at(x, y, thetas[i], O.CONSTANT, O.SQEXP)
        rec["likelihood"].append(lk if math.isfinite(lk) else None)
        rec["status"].append(int(st))
        out[key] = rec
        print(f"sweep {i + 1}/{len(thetas)}: status {st} lkh {lk!r} ({time.time() - t0:.0f}s)", flush=True)
        yield  # checkpoint after every candidate


def part_expert(out):
    n, d = 8192, 16
    x = O.lhs_classic(n, d, 7)
    y = O.griewank(x)
    theta = np.full(d, 0.5 / math.sqrt(d))
    xq = queries(100000, d, 7)[:1000]  # the first 1000 rows of config 5's 100 000 query points
    t0 = time.time()
    gp = O.fit_fixed(x, y, theta, O.CONSTANT, O.SQEXP)
    out[f"expert_n{n}_d{d}"] = {
        "n": n, "d": d, "seed": 7, "corr": O.SQEXP, "theta": theta.tolist(), "likelihood": gp.likelihood,
        "sigma2": gp.inner.sigma2, "min_pivot": float(np.diag(gp.inner.r_chol).min()),
        "query": "np.random.default_rng(7).random((100000, 16))[:1000]",
        "predict": gp.predict(xq).tolist(), "predict_var": gp.predict_var(xq).tolist()}
    print(f"expert: lkh {gp.likelihood!r} ({time.time() - t0:.0f}s)", flush=True)


def part_experts(out):
    """config 5, experts 1..7 (seeds 8..14) on the first 100 query points, and the mixture (smooth / hard recombination,
    crates/moe/src/algorithm.rs:670-685, 894-910) of all eight ORACLE experts on those points, with the responsibilities
    of oracle/moe_oracle.py (weights / means / covariances drawn as in tests/test_gpu_configs.py, seed 5)."""
    from oracle import moe_oracle as MO
    n, d, k, mq = 8192, 16, 8, 100
    theta = np.full(d, 0.5 / math.sqrt(d))
    xq = queries(100000, d, 7)[:mq]
    e0 = out["expert_n8192_d16"]
    per_y, per_v = [np.array(e0["predict"][:mq])], [np.array(e0["predict_var"][:mq])]
    experts = {}
    for e in range(1, k):
        x = O.lhs_classic(n, d, 7 + e)
        y = O.griewank(x)
        t0 = time.time()
        gp = O.fit_fixed(x, y, theta, O.CONSTANT, O.SQEXP)
        yp, vp = gp.predict(xq), gp.predict_var(xq)
        experts[str(e)] = {"seed": 7 + e, "likelihood": gp.likelihood, "sigma2": gp.inner.sigma2,
                           "min_pivot": float(np.diag(gp.inner.r_chol).min()), "predict": yp.tolist(),
                           "predict_var": vp.tolist()}
        per_y.append(yp)
        per_v.append(vp)
        print(f"expert {e}: lkh {gp.likelihood!r} ({time.time() - t0:.0f}s)", flush=True)
    per_y, per_v = np.array(per_y), np.array(per_v)
    rng = np.random.default_rng(5)
    w = rng.random(k) + 0.5
    w /= w.sum()
    means = rng.random((k, d))
    covs = np.array([np.eye(d) * 0.3] * k)
    gmo = MO.GaussianMixtureOracle(w, means, covs, 0.9)
    p = gmo.predict_probas(xq)
    c = gmo.predict(xq)
    rows = np.arange(mq)
    out["experts_1_to_7_n8192_d16"] = {
        "n": n, "d": d, "theta": theta.tolist(), "corr": O.SQEXP, "m": mq,
        "query": "np.random.default_rng(7).random((100000, 16))[:100]", "experts": experts,
        "mixture": {"gmm": "rng = default_rng(5); w = rng.random(8) + 0.5; w /= w.sum(); means = rng.random((8, 16)); "
                           "covs = 0.3 I; heaviside factor 0.9",
                    "clusters": c.tolist(),
                    "smooth_predict": (per_y * p.T).sum(axis=0).tolist(),
                    "smooth_predict_var": (per_v * p.T * p.T).sum(axis=0).tolist(),
                    "hard_predict": per_y[c, rows].tolist(), "hard_predict_var": per_v[c, rows].tolist()}}


def part_xgrad300(out):
    """x-gradients at d = 300 against the oracle itself (not finite differences of the library's own predictions): the training
    set, theta and query points of tests/test_gpu_parity.py::test_x_gradients_beyond_256_dimensions, rows 0 and 1."""
    ms = _multistart()
    n, d = 200, 300
    x = O.lhs_classic(n, d, 5)   # == egobox_amd.workload.make_training_set(n, d, seed=5)
    y = O.griewank(x)
    xq = x.min(0) + (x.max(0) - x.min(0)) * np.random.default_rng(1).random((130, d))
    for corr_id, corr in ((0, O.SQEXP), (3, O.MATERN52)):
        theta = np.full(d, 0.5 / math.sqrt(d)) * (2.0 if corr_id == 0 else 1.0)
        t0 = time.time()
        gp = O.fit_fixed(x, y, theta, O.CONSTANT, corr)
        gy, gv = gp.predict_valvar_gradients(xq[:2])
        out[f"xgrad_n{n}_d{d}_{corr}"] = {
            "n": n, "d": d, "seed": 5, "corr": corr, "corr_id": corr_id, "theta": theta.tolist(),
            "query": "x.min(0) + (x.max(0) - x.min(0)) * default_rng(1).random((130, 300)), rows 0 and 1",
            "likelihood": gp.likelihood, "predict_gradients": gy.tolist(), "predict_var_gradients": gv.tolist()}
        print(f"xgrad300 {corr}: |gy| {np.linalg.norm(gy):.6e} |gv| {np.linalg.norm(gv):.6e} ({time.time() - t0:.0f}s)", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "large_n.json"))
    args = ap.parse_args()
    out = {}
    if os.path.exists(args.out):
        with open(args.out) as f:
            out = json.load(f)

    def save():
        out["_generated_by"] = "tests/golden/make_large_n.py (oracle/gp_oracle.py, numpy %s)" % np.__version__
        tmp = args.out + ".tmp"
        with open(tmp, "w") as f:
            json.dump(out, f)
        os.replace(tmp, args.out)

    want = lambda p: args.only in ("", p)  # noqa: E731
    if want("expert"):
        part_expert(out)
        save()
    if args.only == "experts" or (args.only == "" and "experts_1_to_7_n8192_d16" not in out):
        part_experts(out)
        save()
    if want("grad"):
        part_grad(out)
        save()
    if args.only == "grad16384" or (args.only == "" and f"grad_n16384_d32_{O.MATERN52}" not in out):
        part_grad16384(out)
        save()
    if args.only == "xgrad300" or (args.only == "" and f"xgrad_n200_d300_{O.SQEXP}" not in out):
        part_xgrad300(out)
        save()
    if want("fit"):
        for _ in part_fit(out):
            save()
        save()
    if want("sweep"):
        for _ in part_sweep(out):
            save()
        save()


if __name__ == "__main__":
    main()
