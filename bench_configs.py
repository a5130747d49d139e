#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs (bench.py stays the driver's contract).

  config 2  n=4096  d=8   sq-exp      fixed-theta fits/s, Cholesky TFLOP/s, K1 GB/s
  config 3  n=16384 d=32  Matern-5/2  likelihood evals/s, one likelihood+gradient evaluation
  config 4  theta sweep (n=16384 d=32 sq-exp): evals/s of egx_gp_likelihood_batch on this GPU
            (the 8-GPU run shards candidates k mod G and all-gathers, egobox_amd/sweep.py)
  config 5  one expert n=8192 d=16: predict / predict_var on m=100000 query points

Prints one JSON object per config.  Run on the GPU box:  python bench_configs.py [--quick]
"""
import argparse
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))


def timeit(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", type=int, default=0)
    args = ap.parse_args()
    import egobox_amd as egx
    from egobox_amd import workload
    from concurrent.futures import ThreadPoolExecutor

    def emit(d):
        print(json.dumps(d), flush=True)

    # ---------------- config 2
    if args.only in (0, 2):
        n, d = 4096, 8
        x, y = workload.make_training_set(n, d, 42)
        th = workload.default_theta(d)
        hs = [egx.GpHandle(x, y) for _ in range(8)]
        pool = ThreadPoolExecutor(8)
        t1 = timeit(lambda: hs[0].finalize(th), 10)
        tm = hs[0].timings()
        tk = {k: timeit(lambda: list(pool.map(lambda h: h.finalize(th), hs[:k])), 10) / k for k in (2, 4, 8)}
        # the throughput shape of the product: ONE handle, twelve candidates factored in lock-step by one launch sequence
        # (what a tuned fit's round of COBYLA trial points is), 48 candidates per measurement
        hb = egx.GpHandle(x, y, n_workspaces=12)
        ths = np.stack([th * (1.0 + 0.01 * c) for c in range(48)])
        tb = timeit(lambda: hb.likelihood_batch(ths), 5) / len(ths)
        hb.close()
        emit({"config": 2, "n": n, "d": d, "corr": "SquaredExponential", "fits_per_s_1_in_flight": 1 / t1,
              "likelihoods_per_s_lockstep_12_one_handle": 1 / tb,
              "cholesky_tflops_lockstep_12": n ** 3 / 3 / tb / 1e12,
              "fits_per_s_2_in_flight": 1 / tk[2], "fits_per_s_4_in_flight": 1 / tk[4],
              "fits_per_s_8_in_flight": 1 / tk[8], "potrf_ms": tm["potrf_ms"],
              "cholesky_tflops": tm["potrf_flops"] / tm["potrf_ms"] / 1e9, "corr_build_ms": tm["corr_build_ms"],
              "corr_build_gbps": tm["corr_bytes"] / tm["corr_build_ms"] / 1e6})
        for h in hs:
            h.close()

    # ---------------- config 2b: a TUNED fit (what COBYLA multiplies): default n_start = 10, max_eval = 50
    if args.only in (0, 2, 7):
        for n, d in ((4096, 8), (1024, 8), (256, 4)):
            x, y = workload.make_training_set(n, d, 42)
            out = {"config": "2-tuned-fit", "n": n, "d": d, "corr": "SquaredExponential", "n_start": 10, "max_eval": 50}
            for opt, nws in (("cobyla", 1), ("cobyla", 4), ("cobyla", 11), ("lbfgs", 1), ("lbfgs", 11)):
                for rep in range(2):  # the second run finds its resources in the library's pool
                    t0 = time.perf_counter()
                    gp = egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr()) \
                        .n_start(10).max_eval(50).optimizer(opt).n_workspaces(nws).fit(x, y)
                    dt = time.perf_counter() - t0
                    key = f"{opt}_{nws}ws"
                    out[key] = {"wall_s": dt, "likelihood_evals": int(gp.n_evals), "evals_per_s": gp.n_evals / dt,
                                "likelihood": float(gp.likelihood())}
                    gp.close()
            out["note"] = ("ThetaTuning::Full default: 11 starts x clamp(10 d, 25, max_eval) evaluations "
                           "(algorithm.rs:928-945); cobyla = Powell's method restated (csrc/cobyla.h, all starts in "
                           "lock-step through one likelihood batch per round), lbfgs = an extension on the new gradient")
            emit(out)

    # ---------------- config 3
    if args.only in (0, 3):
        n, d = 16384, 32
        x, y = workload.make_training_set(n, d, 42)
        th = workload.default_theta(d)
        # one evaluation in flight (stage times), sixteen in flight as two lock-step groups of eight (throughput), and the
        # likelihood + theta-gradient alone and in a lock-step batch of eight
        h = egx.GpHandle(x, y, corr=3, n_workspaces=1)
        h.finalize(th)
        t1 = timeit(lambda: h.finalize(th * 1.01), 3)
        tm = h.timings()
        h.likelihood_grad(th)
        t0 = time.perf_counter()
        lk, g, st = h.likelihood_grad(th * 1.001)
        tg = time.perf_counter() - t0
        h.close()
        h = egx.GpHandle(x, y, corr=3, n_workspaces=16)
        thetas = np.stack([th * (1 + 0.003 * i) for i in range(48)])
        h.likelihood_batch(thetas[:16])
        t = timeit(lambda: h.likelihood_batch(thetas), 1) / len(thetas)
        h.close()
        h = egx.GpHandle(x, y, corr=3, n_workspaces=8)
        h.set_lockstep(8)
        h.likelihood_grad_batch(thetas[:8])
        t0 = time.perf_counter()
        lkb, gb, stb = h.likelihood_grad_batch(thetas[8:16])
        tgb = (time.perf_counter() - t0) / 8
        h.close()
        emit({"config": 3, "n": n, "d": d, "corr": "Matern52", "fixed_theta_fit_ms_one_in_flight": t1 * 1e3,
              "likelihood_evals_per_s_16_in_flight": 1 / t,
              "corr_build_ms": tm["corr_build_ms"], "corr_build_gbps": tm["corr_bytes"] / tm["corr_build_ms"] / 1e6,
              "potrf_ms": tm["potrf_ms"], "cholesky_tflops": tm["potrf_flops"] / tm["potrf_ms"] / 1e9,
              "likelihood_plus_gradient_s": tg, "likelihood_plus_gradient_frac_of_fp64_peak": float(n) ** 3 / tg / 78.6e12,
              "likelihood_plus_gradient_s_per_candidate_lockstep_8": tgb,
              "likelihood_plus_gradient_frac_of_fp64_peak_lockstep_8": float(n) ** 3 / tgb / 78.6e12,
              "grad_status": int(st), "grad_norm": float(np.linalg.norm(g)), "likelihood": lk})

    # ---------------- config 4
    if args.only in (0, 4):
        n, d = 16384, 32
        x, y = workload.make_training_set(n, d, 42)
        k = 32 if args.quick else 512
        thetas = egx.theta_sweep_candidates(512, d)[:k]
        # through the boundary's multi-GPU entry point (egx_sweep_*, one-rank RCCL communicator on this box), in the
        # product's shape: 16 candidates in flight as two lock-step groups of eight (left-looking)
        h = egx.Sweep(x, y, rank=0, world=1, id_bytes="new", n_workspaces=16)
        h.likelihood(thetas[:16])
        t0 = time.perf_counter()
        lk, st = h.likelihood(thetas)
        t = time.perf_counter() - t0
        emit({"config": 4, "n": n, "d": d, "candidates_timed": k, "of_sweep": 512, "evals_per_s_this_gpu": k / t,
              "status_counts": {int(s): int((st == s).sum()) for s in np.unique(st)},
              "best": float(np.max(lk[st == 0])) if (st == 0).any() else None,
              "projected_512_evals_s_1gpu": 512 * t / k, "projected_512_evals_s_8gpu": 512 * t / k / 8})
        h.close()

    # ---------------- config 5
    if args.only in (0, 5):
        n, d, m = 8192, 16, (20000 if args.quick else 100000)
        x, y = workload.make_training_set(n, d, 7)
        th = workload.default_theta(d)
        h = egx.GpHandle(x, y)
        t0 = time.perf_counter()
        h.finalize(th)
        tfit = time.perf_counter() - t0
        xq = np.random.default_rng(7).random((m, d))
        h.predict(xq[:1000])
        t0 = time.perf_counter()
        yp = h.predict(xq)
        tp = time.perf_counter() - t0
        h.predict_var(xq[:1000])
        t0 = time.perf_counter()
        vp = h.predict_var(xq)
        tv = time.perf_counter() - t0
        h.predict_gradients(xq[:1000])
        t0 = time.perf_counter()
        h.predict_gradients(xq)
        tgy = time.perf_counter() - t0
        h.predict_var_gradients(xq[:1000])
        mg = m // 4
        t0 = time.perf_counter()
        h.predict_var_gradients(xq[:mg])
        tgv = time.perf_counter() - t0
        t1 = timeit(lambda: h.predict_valvar_gradients(xq[:1]), 20)
        emit({"config": "5-x-gradients", "n": n, "d": d, "predict_gradients_points_per_s": m / tgy,
              "predict_var_gradients_points_per_s": mg / tgv, "m_var_gradients": mg,
              "one_point_valvar_gradients_ms": t1 * 1e3,
              "note": "batched; the reference evaluates point by point with two n^2 triangular solves plus "
                      "R^-1 F and chol(F^T R^-1 F) per point (algorithm.rs:555-617)"})
        emit({"config": 5, "n": n, "d": d, "m": m, "fit_s": tfit, "predict_points_per_s": m / tp,
              "predict_var_points_per_s": m / tv, "predict_var_trsm_tflops": (float(n) * n * m) / tv / 1e12,
              "var_min": float(vp.min()), "var_max": float(vp.max()), "y_mean": float(yp.mean()),
              "projected_8_experts_8gpu_points_per_s": m / tv})
        h.close()

    # ---------------- config 5b: the whole mixture on ONE GPU (8 experts resident), smooth and hard recombination
    if args.only in (0, 6):
        from egobox_amd.moe import GaussianMixture, GpMixture
        k, n, d, m = 8, 8192, 16, (20000 if args.quick else 100000)
        rng = np.random.default_rng(11)
        experts, means = [], []
        t0 = time.perf_counter()
        for e in range(k):
            x = workload.lhs(n, d, 100 + e)
            y = workload.griewank(x) * (1.0 + 0.1 * e)
            experts.append(egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr())
                           .theta_tuning(egx.ThetaTuning.Fixed(workload.default_theta(d))).fit(x, y))
            means.append(rng.random(d))
        tfit = time.perf_counter() - t0
        gmx = GaussianMixture(np.full(k, 1.0 / k), np.array(means), np.array([np.eye(d) * 0.05] * k))
        xq = rng.random((m, d))
        out = {"config": "5-mixture-1gpu", "experts": k, "n_per_expert": n, "d": d, "m": m, "fit_all_experts_s": tfit}
        for recomb in ("smooth", "hard"):
            mix = GpMixture(experts, gmx, recomb)
            mix.predict_var(xq[:2000])
            t0 = time.perf_counter()
            v = mix.predict_var(xq)
            t = time.perf_counter() - t0
            out[f"predict_var_{recomb}_points_per_s"] = m / t
            out[f"predict_var_{recomb}_checksum"] = float(v.sum())
        out["note"] = ("smooth = every expert sees all points (8 x n^2 m flops); hard = points routed once, ONE batched "
                       "call per expert (the reference calls the expert once per row, crates/moe/src/algorithm.rs:894-910)")
        emit(out)
        for e in experts:
            e.close()

    if args.only in (0, 8):
        sgp_bench(emit, args.quick)


def sgp_bench(emit, quick):
    """Sparse GP: one FITC / VFE likelihood evaluation = R_nz (ALU/HBM) + trsm (n nz^2 MFMA) +
    split-K Gram product (n nz^2 MFMA) + two nz^3/3 factorisations."""
    import egobox_amd as egx
    for n, nz, d in ((200_000, 256, 8),) if quick else ((200_000, 256, 8), (1_000_000, 512, 8)):
        rng = np.random.default_rng(3)
        x = rng.random((n, d)) * 2 - 1
        y = np.sin(3 * x[:, 0]) + 0.5 * np.cos(2 * x[:, 1]) * x[:, 2] + 0.1 * rng.standard_normal(n)
        z = x[rng.permutation(n)[:nz]].copy()
        theta = np.full(d, 0.8)
        out = {"config": "sgp", "n": n, "nz": nz, "d": d}
        for method, name in ((0, "fitc"), (1, "vfe")):
            h = egx.SgpHandle(x, y, z, corr=0, method=method)
            t = timeit(lambda: h.likelihood(theta, 1.0, 0.01), 5)
            out[f"{name}_likelihood_ms"] = t * 1e3
            out[f"{name}_mfma_tflops"] = 4.0 * n * nz * nz / 2 / t / 1e12  # trsm n nz^2 / 2 MACs + Gram n nz^2 / 2 MACs, x 2 flop
            h.finalize(theta, 1.0, 0.01)
            xq = rng.random((200_000, d)) * 2 - 1
            h.predict(xq[:1000])
            t0 = time.perf_counter()
            h.predict(xq)
            out[f"{name}_predict_points_per_s"] = xq.shape[0] / (time.perf_counter() - t0)
            h.predict_var(xq[:1000])
            t0 = time.perf_counter()
            h.predict_var(xq)
            out[f"{name}_predict_var_points_per_s"] = xq.shape[0] / (time.perf_counter() - t0)
            h.close()
        emit(out)


if __name__ == "__main__":
    main()
