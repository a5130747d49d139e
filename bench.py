#!/usr/bin/env python3
"""Headline benchmark: fixed-theta GP fits per second, n = 16384, d = 32, squared exponential.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
     or plain `python bench.py --gpus N`, which re-launches itself under torch.distributed.run with N ranks on
     127.0.0.1; either way the line is only printed when the world size IS N and the RCCL communicator inside the
     library has N ranks)

One "step" = one pass of the hot path over one batch of `--sweep-batch` (default 192 = three lock-step groups of eight per
step on each of 8 GPUs) candidate thetas of a theta sweep: every candidate is one fit in north_star's sense --
correlation-matrix build (K1) + blocked FP64-MFMA Cholesky with fused forward solves (K3/K4) + GLS / reduced
likelihood, i.e. one evaluation of the objective the reference's COBYLA multiplies
(crates/gp/src/algorithm.rs:880-897, 988-1056).  The batch is FIXED as N grows (strong scaling): rank r evaluates
candidates r, r + N, ... through `egx_sweep_likelihood` (include/egx_gp.h), whose RCCL all-gather of the
(likelihood, status) pairs runs inside libegx_gp_hip.so, inside the timed region, every step -- also at N = 1
(one-rank communicator), so the measured code path is the same at every N.  On every GPU the candidates are factored
in LOCK-STEP groups (egx_gp_set_lockstep: one launch sequence per group of eight, grid.z = candidate; a handle of this
size factors left-looking over its panel groups, kernels_chol.hip launch_potrf), two groups in flight.  The training set is resident in HBM before the timed region starts.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the Cholesky trailing update, FP64 MFMA
bound), measured in a separate leg with ONE fit in flight; `cpu_baseline` is the blas-feature-shaped CPU path
(oracle/cpu_baseline.py: OpenMP correlation build + LAPACK dpotrf) MEASURED at the full size in its own process on the
GPU box's host -- one fit at its best thread count (latency mode) -- and `cpu_baseline_concurrent` as many such fits
side by side as the host holds (throughput mode, the shape of the reference's rayon multistart); the speed-ups are
labelled by mode.  `other_configs`: bounded side measurements of BASELINE configs 3 and 5 and of d = 64.
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# public MI355X figures (the CDNA4 guide in this image lists no FP64 matrix peak):
FP64_MFMA_PEAK_TFLOPS = 78.6
HBM_PEAK_GBPS = 8000.0
PMC_SUMMARY = next((p for p in (os.path.join(ROOT, "profiles", "r03_pmc_update_kernel.json"),
                                os.path.join(ROOT, "profiles", "r02_pmc_update_kernel.json")) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", "r02_pmc_update_kernel.json"))


def cpu_baseline(n, d, timeout_s=900):
    """oracle/cpu_baseline.py in a FRESH process (no torch / second OpenBLAS threadpool beside LAPACK),
    at the full problem size.  Returns its JSON record."""
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        env.pop(k, None)  # all host cores
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--n", str(n), "--d", str(d), "--blas-threads", "16,32,64,128"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


def cpu_baseline_concurrent(n, d, threads_per_fit, n_procs=4, timeout_s=600):
    """The like-for-like CPU figure for a THROUGHPUT number: the reference's multistart is rayon-parallel over starts
    (crates/gp/src/algorithm.rs:928-945), i.e. several fits side by side, each with a slice of the host's cores.  As
    many oracle/cpu_baseline.py processes as fit the host at `threads_per_fit` threads each (the setting a single fit
    is fastest with) start their fit at the same instant; value = processes / (last end - common start)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # (r03 run 4 on the 2 x 64-core / 256-thread GPU box: 16 processes x 16 threads took 69 s for 16 fits -- 0.23 fits/s,
    #  against 0.55 for ONE such process alone: OpenBLAS' spinning threads and the host's memory bandwidth do not carry
    #  sixteen 2 GiB factorisations.  Four side by side is the bounded probe the default run can afford.)
    procs = max(1, min(n_procs, cores // max(1, 2 * threads_per_fit)))
    start_at = time.time() + 12.0 + 0.25 * procs  # imports + LHS / Griewank generation + normalisation happen before it
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(threads_per_fit)
    env["OPENBLAS_NUM_THREADS"] = str(threads_per_fit)
    cmd = [sys.executable, "-m", "oracle.cpu_baseline", "--n", str(n), "--d", str(d), "--blas-threads",
           str(threads_per_fit), "--start-at", repr(start_at)]
    ps = [subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
          for _ in range(procs)]
    recs = []
    for p in ps:
        try:
            so, se = p.communicate(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            p.kill()
            return {"error": "timeout"}
        if p.returncode != 0:
            return {"error": se[-300:]}
        recs.append(json.loads(so.strip().splitlines()[-1]))
    late = max(r["wall_start"] for r in recs) - start_at
    wall = max(r["wall_end"] for r in recs) - min(r["wall_start"] for r in recs)
    return {"value": procs / wall, "unit": "fits/s", "cores": procs * threads_per_fit, "kind": "port",
            "processes": procs, "threads_per_process": threads_per_fit, "host_threads": cores,
            "wall_s_all_fits": wall, "fit_s_each": [r["seconds"]["fit"] for r in recs],
            "start_skew_s": late,
            "sample": (f"{procs} simultaneous fixed-theta fits at n={n} d={d}, one process each with {threads_per_fit} "
                       f"OpenMP / OpenBLAS threads ({procs * threads_per_fit} of the host's {cores} hardware threads), common "
                       f"start: {procs} fits in {wall:.2f} s")}


def cpu_baseline_reference_shaped(n_full, d):
    """SURVEY 8d baseline (i): the C restatement in the shape of the reference's DEFAULT build (single thread,
    materialised difference table, unblocked Cholesky; oracle/ref_shaped.c) on two bounded samples, extrapolated
    with a fitted a n^2 + b n^3.  Reported next to `cpu_baseline`, never the target."""
    from oracle import gp_oracle as O
    from oracle import ref_shaped
    if not ref_shaped.available():
        return None
    ts, ns = [], (1024, 2048)
    for n_s in ns:
        x = O.lhs_classic(n_s, d, 42)
        y = O.griewank(x)
        t0 = time.perf_counter()
        r = ref_shaped.likelihood(x, y, np.full(d, 0.5 / np.sqrt(d)))
        ts.append(time.perf_counter() - t0)
    a, b = np.linalg.solve(np.array([[ns[0] ** 2, ns[0] ** 3], [ns[1] ** 2, ns[1] ** 3]], dtype=float), np.array(ts))
    a, b = max(a, 0.0), max(b, 0.0)
    t_full = a * n_full ** 2 + b * n_full ** 3
    return {"value": 1.0 / t_full, "unit": "fits/s", "cores": 1, "kind": "port",
            "sample": (f"oracle/ref_shaped.c (single thread, (pairs,d) table + unblocked Cholesky) at n={ns[0]} "
                       f"({ts[0]:.2f}s) and n={ns[1]} ({ts[1]:.2f}s), d={d}; a n^2 + b n^3 EXTRAPOLATED to n={n_full} "
                       f"({t_full:.0f}s per fit; the table alone would need {n_full * (n_full - 1) // 2 * d * 8 / 1e9:.1f} GB)"),
            "sample_seconds": float(sum(ts)), "sample_status": int(r["status"])}


def vendor_yardstick(torch, gpu, sizes=(16384, 4096)):
    """The same-GPU external anchor (BASELINE.json publishes no number): torch.linalg.cholesky, i.e. the vendor's
    rocSOLVER / hipSOLVER FP64 potrf behind PyTorch-ROCm, on a well-conditioned SPD matrix of the metric's size.  Run
    AFTER the timed region, never part of the product path.  TFLOP/s = n^3/3 over the best of three calls."""
    res = {}
    dev = torch.device(f"cuda:{gpu}")
    for n in sizes:
        try:
            gen = torch.Generator(device=dev).manual_seed(1)
            a = torch.randn(n, 256, dtype=torch.float64, device=dev, generator=gen)
            spd = a @ a.T
            spd.diagonal().add_(float(n))
            del a
            torch.linalg.cholesky(spd)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                torch.linalg.cholesky(spd)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            del spd
            torch.cuda.empty_cache()
            res[f"n{n}"] = {"ms": min(ts) * 1e3, "tflops": n ** 3 / 3 / min(ts) / 1e12,
                            "frac_of_fp64_peak": n ** 3 / 3 / min(ts) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
        except Exception as e:  # noqa: BLE001 - a side figure
            res[f"n{n}"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    res["what"] = (f"torch.linalg.cholesky (torch {torch.__version__}: rocSOLVER / hipSOLVER potrf), float64, lower, one "
                   f"matrix, out-of-place, best of 3; a yardstick on the same GPU, not part of the product")
    return res


def host_fp64_peak():
    """cores x 16 flop/clk x clock from lscpu (two 256-bit FMA pipes per core): the denominator the CPU baseline's dpotrf
    rate is quoted against."""
    try:
        txt = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
    except Exception:  # noqa: BLE001
        return None
    f = {}
    for ln in txt.splitlines():
        if ":" in ln:
            k, v = ln.split(":", 1)
            f[k.strip()] = v.strip()
    try:
        sockets = int(f.get("Socket(s)", "1"))
        cps = int(f.get("Core(s) per socket", "0"))
        mhz = float(f.get("CPU max MHz", f.get("CPU MHz", "0")).split()[0])
        if cps <= 0 or mhz <= 0:
            return None
        cores = sockets * cps
        return {"model": f.get("Model name"), "sockets": sockets, "cores": cores, "threads": int(f.get("CPU(s)", "0")),
                "clock_mhz_max": mhz, "flop_per_clk_per_core_assumed": 16,
                "peak_gflops": cores * 16 * mhz / 1e3}
    except Exception:  # noqa: BLE001
        return None


def stream_kernel_signature():
    """sha256 of the source text of the k_gemm_stream template (the kernel `roofline` describes): the committed PMC summary
    carries the hash it was measured with."""
    import hashlib
    try:
        with open(os.path.join(ROOT, "egobox_amd", "csrc", "kernels_chol.hip")) as f:
            txt = f.read()
        a = txt.rfind("template", 0, txt.index("void k_gemm_stream("))
        return hashlib.sha256(txt[a:txt.index("\n}\n", a) + 3].encode()).hexdigest()
    except (OSError, ValueError):
        return "unknown"


def measured_traffic(n, d):
    """HBM/fabric bytes per launch of the update kernel from the committed PMC summary (separate --pmc passes of
    `bench.py --steps 1`), corrected as the MI355X guide prescribes: on gfx950 FETCH_SIZE tallies a 16 B/lane
    streaming read at half its bytes.  The kernel's 16 B/lane reads are the A/B panel tiles; its C tile read is
    8 B/lane, so:  traffic = WRITE + C_read + 2 * (FETCH - C_read)."""
    if (n, d) != (16384, 32) or not os.path.exists(PMC_SUMMARY):
        return None, None
    with open(PMC_SUMMARY) as f:
        p = json.load(f)
    fetch, write, c_read = p["fetch_bytes_per_launch"], p["write_bytes_per_launch"], p["c_read_bytes_per_launch"]
    return write + c_read + 2.0 * max(0.0, fetch - c_read), p


def size_ladder(egx, workload, gpu):
    """The Cholesky of ONE fixed-theta fit across sizes, driver-visible: n in {2048, 4096, 8192, 16384} x {one evaluation in
    flight on a one-workspace handle, a lock-step batch on a handle with as many workspaces}: milliseconds per
    factorisation, TFLOP/s for n^3 / 3 flop, fraction of the FP64 peak, and the schedule the handle reports
    (egx_gp_get_schedule).  Squared exponential, the metric's data family.  ~10 s."""
    out = {}
    for n, d, width in ((2048, 8, 12), (4096, 8, 12), (8192, 16, 12), (16384, 32, 8)):
        x, y = workload.make_training_set(n, d, seed=42)
        th = workload.default_theta(d) * (3.0 if d <= 8 else 1.0)
        flop = float(n) ** 3 / 3.0
        h = egx.GpHandle(x, y, mean=0, corr=0, device=gpu, n_workspaces=1)
        h.finalize(th)
        reps = 10 if n <= 4096 else 3
        t0 = time.perf_counter()
        pm = []
        for j in range(reps):
            h.finalize(th * (1.0 + 1e-3 * j))
            pm.append(h.timings()["potrf_ms"])
        t_fit = (time.perf_counter() - t0) / reps
        sched = h.schedule()
        h.close()
        one = {"fit_ms": t_fit * 1e3, "potrf_ms": float(np.median(pm)), "potrf_tflops": flop / float(np.median(pm)) / 1e9,
               "potrf_frac_of_fp64_peak": flop / float(np.median(pm)) / 1e9 / FP64_MFMA_PEAK_TFLOPS, "schedule": sched}
        h = egx.GpHandle(x, y, mean=0, corr=0, device=gpu, n_workspaces=width)
        h.set_lockstep(width)
        ths = np.stack([th * (1.0 + 0.004 * c) for c in range(2 * width)])
        h.likelihood_batch(ths[:width])
        t0 = time.perf_counter()
        lk, st = h.likelihood_batch(ths * 1.001)
        t_b = (time.perf_counter() - t0) / (2 * width)
        schedb = h.schedule()
        h.close()
        out[f"n{n}_d{d}"] = {
            "one_in_flight": one,
            f"lockstep_{width}": {"evaluation_ms_per_candidate": t_b * 1e3, "likelihoods_per_s": 1.0 / t_b,
                                   "tflops_for_n3_over_3": flop / t_b / 1e12,
                                   "frac_of_fp64_peak": flop / t_b / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                   "statuses_ok": int(np.sum(st == 0)), "schedule": schedb}}
    return out


def other_configs(egx, workload, gpu):
    """Bounded side measurements in the SAME driver-run line (never `value`): BASELINE config 3 (Matern-5/2 likelihood
    and likelihood + theta-gradient at n = 16384, d = 32), config 5 (one expert n = 8192, d = 16: predict / predict_var on
    100 000 points), and the metric's shape at d = 64 (north_star: "d up to 64").  ~10 s in all."""
    res = {}
    n, d = 16384, 32
    x, y = workload.make_training_set(n, d, seed=42)
    th = workload.default_theta(d)
    h = egx.GpHandle(x, y, mean=0, corr=3, device=gpu, n_workspaces=1)
    h.finalize(th)
    t0 = time.perf_counter()
    h.finalize(th * 1.01)
    t_fit = time.perf_counter() - t0
    tm = h.timings()
    h.likelihood_grad(th * 0.99)  # (first call: allocates the C^-T scratch)
    t_gs = []
    for j in range(2):
        t0 = time.perf_counter()
        lk, g, st = h.likelihood_grad(th * (1.0 + 0.005 * j))
        t_gs.append(time.perf_counter() - t0)
    t_grad = min(t_gs)
    h.close()
    # likelihood + gradient = Cholesky n^3/3 + C^-T n^3/3 + R^-1 = C^-T C^-1 n^3/3: n^3 flop on the FP64 MFMA pipe
    gflop = float(n) ** 3
    grad_roof = {"bound": "mfma", "flops": gflop, "achieved": gflop / t_grad / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS,
                 "unit": "TFLOP/s", "frac": gflop / t_grad / 1e12 / FP64_MFMA_PEAK_TFLOPS, "ms": t_grad * 1e3,
                 "how": "egx_gp_likelihood_grad, ONE candidate in flight, host wall time of the call (correlation build, "
                        "Cholesky, host GLS, C^-T, gamma, R^-1, trace kernel, read-back) against n^3 flop"}
    gb = None
    try:  # the same through egx_gp_likelihood_grad_batch: 8 candidates as one lock-step group (32 GiB of M + C^-T)
        hb = egx.GpHandle(x, y, mean=0, corr=3, device=gpu, n_workspaces=8)
        hb.set_lockstep(8)
        ths = np.stack([th * (1.0 + 0.004 * c) for c in range(8)])
        hb.likelihood_grad_batch(ths)
        t0 = time.perf_counter()
        lkb, gbv, stb = hb.likelihood_grad_batch(ths * 1.001)
        t_b = time.perf_counter() - t0
        hb.close()
        gb = {"candidates": 8, "ms_per_candidate": t_b / 8 * 1e3, "achieved": 8 * gflop / t_b / 1e12,
              "frac": 8 * gflop / t_b / 1e12 / FP64_MFMA_PEAK_TFLOPS, "statuses_ok": int(np.sum(stb == 0))}
        # ... and 16 candidates as two lock-step groups of eight in flight (the trace kernel and the host half of one
        # group beside the other group's factorisation; 68 GiB of M + C^-T)
        hb = egx.GpHandle(x, y, mean=0, corr=3, device=gpu, n_workspaces=16)
        hb.set_lockstep(8)
        ths = np.stack([th * (1.0 + 0.004 * c) for c in range(16)])
        hb.likelihood_grad_batch(ths)
        t0 = time.perf_counter()
        lkb, gbv, stb = hb.likelihood_grad_batch(ths * 1.001)
        t_b = time.perf_counter() - t0
        hb.close()
        gb["two_groups_in_flight"] = {"candidates": 16, "ms_per_candidate": t_b / 16 * 1e3, "achieved": 16 * gflop / t_b / 1e12,
                                      "frac": 16 * gflop / t_b / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                      "statuses_ok": int(np.sum(stb == 0))}
    except Exception as e:  # noqa: BLE001 - a side figure
        gb = dict(gb or {}, error=f"{type(e).__name__}: {e}"[:200])
    res["config3_matern52_n16384_d32"] = {
        "fixed_theta_fit_ms": t_fit * 1e3, "corr_build_ms": tm["corr_build_ms"],
        "corr_build_gbps": tm["corr_bytes"] / tm["corr_build_ms"] / 1e6, "potrf_ms": tm["potrf_ms"],
        "cholesky_tflops": tm["potrf_flops"] / tm["potrf_ms"] / 1e9, "likelihood_plus_theta_gradient_ms": t_grad * 1e3,
        "gradient_roofline": grad_roof, "gradient_roofline_lockstep_batch_of_8": gb,
        "gradient_status": int(st), "gradient_norm": float(np.linalg.norm(g))}
    n5, d5, m5 = 8192, 16, 100000
    x5, y5 = workload.make_training_set(n5, d5, seed=7)
    h = egx.GpHandle(x5, y5, mean=0, corr=0, device=gpu, n_workspaces=1)
    h.finalize(workload.default_theta(d5))
    xq = np.random.default_rng(7).random((m5, d5))
    h.predict_valvar(xq[:2000])
    t0 = time.perf_counter()
    h.predict(xq)
    t_p = time.perf_counter() - t0
    t0 = time.perf_counter()
    h.predict_var(xq)
    t_v = time.perf_counter() - t0
    h.close()
    res["config5_expert_n8192_d16_m100000"] = {
        "predict_points_per_s": m5 / t_p, "predict_var_points_per_s": m5 / t_v,
        "predict_var_trsm_tflops": float(n5) * n5 * m5 / t_v / 1e12,
        "predict_var_frac_of_fp64_peak": float(n5) * n5 * m5 / t_v / 1e12 / FP64_MFMA_PEAK_TFLOPS}
    # config 5 as a MIXTURE: 8 experts x n = 8192 resident on this GPU, smooth recombination of predict_var on 100 000
    # points INSIDE the library (egx_moe_predict_valvar; with N ranks expert e lives on rank e mod N and the same call
    # carries one all-gather of the partial sums)
    from egobox_amd.moe import GaussianMixture, GpMixture
    k = 8
    rng = np.random.default_rng(5)
    # the experts are fitted the way egobox-moe's expert loop fits them (crates/moe/src/algorithm.rs:167-177: one fixed-theta
    # `fit` per cluster) -- one after the other first (a model per call), then in LOCK-STEP (GpParams.fit_group:
    # egx_gp_create_group + egx_gp_finalize_multi, one launch sequence for the eight factorisations)
    sets5 = [workload.make_training_set(n5, d5, seed=7 + e) for e in range(k)]
    params5 = (egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr())
               .theta_tuning(egx.ThetaTuning.Fixed(workload.default_theta(d5))))
    lone = [params5.fit(*s) for s in sets5[:2]]          # (warm: pool, code objects)
    for e in lone:
        e.close()
    t0 = time.perf_counter()
    lone = [params5.fit(*s) for s in sets5]
    t_lone = time.perf_counter() - t0
    lone_lk = [e.likelihood() for e in lone]
    lone_sched = lone[0].handle.schedule()
    for e in lone:
        e.close()
    xs5, ys5 = np.stack([s[0] for s in sets5]), np.stack([s[1] for s in sets5])
    for e in params5.fit_group(xs5, ys5):                # (warm)
        e.close()
    t0 = time.perf_counter()
    experts = params5.fit_group(xs5, ys5)
    t_group = time.perf_counter() - t0
    hs5 = [e.handle for e in experts]
    th5 = np.tile(workload.default_theta(d5), (k, 1))
    t0 = time.perf_counter()
    egx.finalize_multi(hs5, th5)                          # the factorisations alone (models resident: no create, no upload)
    t_multi = time.perf_counter() - t0
    flop5 = k * float(n5) ** 3 / 3.0
    res["config5_expert_fits_8"] = {
        "experts": k, "n": n5, "d": d5,
        "create_and_fit_one_after_the_other_ms": t_lone * 1e3, "create_and_fit_in_lock_step_ms": t_group * 1e3,
        "refit_in_lock_step_ms": t_multi * 1e3, "refit_tflops_for_n3_over_3": flop5 / t_multi / 1e12,
        "refit_frac_of_fp64_peak": flop5 / t_multi / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        # (a lone handle of this size factors as one flow launch since round 6, a group in lock-step by separate launches:
        #  two schedule rows, the same sums in another order -- the members are no longer the lone fits bit for bit at THIS size)
        "max_rel_diff_to_lone_fits": float(max(abs(e.likelihood() - l) / abs(l) for e, l in zip(experts, lone_lk))),
        "lone_fit_schedule": _schedule_word(lone_sched),
        "note": "egx_gp_create_group + egx_gp_finalize_multi: eight models of one shape, one launch sequence"}
    # the same eight experts TUNED (ThetaTuning::Full: 4 starts x 25 COBYLA evaluations each; crates/moe/src/algorithm.rs:209-262 ->
    # crates/gp/src/algorithm.rs:921-945): egx_gp_fit_multi -- all members' machines in lock-step, a round's trial points as
    # lock-step launch sequences across the group -- against eight egx_gp_fit calls one after the other on one-workspace handles
    # (bit-identical results: tests/test_gpu_pipe.py) and on the twelve-workspace handles GpParams.fit uses
    try:
        lo5, hi5 = np.full(d5, 0.02), np.full(d5, 2.0)
        starts5 = workload.default_theta(d5) * 10.0 ** np.random.default_rng(5).uniform(-0.3, 0.3, size=(4, d5))
        t0 = time.perf_counter()
        ne_multi = egx.fit_multi(hs5, np.tile(starts5, (k, 1, 1)), lo5, hi5, 25)
        t_fm = time.perf_counter() - t0
        lk_multi = [h_.fitted_scalars()[0] for h_ in hs5]
        t_lone1, ne_lone1, same = 0.0, 0, True
        for (x_, y_), lkm in zip(sets5, lk_multi):
            with egx.GpHandle(x_, y_, device=gpu, n_workspaces=1) as h1:
                t0 = time.perf_counter()
                ne_lone1 += int(h1.fit(starts5, lo5, hi5, 25))
                t_lone1 += time.perf_counter() - t0
                same = same and (h1.fitted_scalars()[0] == lkm or h1.schedule()["flow"] == 1)
        t_lone4, ne_lone4 = 0.0, 0
        for x_, y_ in sets5[:2]:
            with egx.GpHandle(x_, y_, device=gpu, n_workspaces=4) as h4:
                t0 = time.perf_counter()
                ne_lone4 += int(h4.fit(starts5, lo5, hi5, 25))
                t_lone4 += time.perf_counter() - t0
        res["config5_expert_tuned_fits_8"] = {
            "experts": k, "starts": 4, "evaluations": int(ne_multi.sum()), "fit_multi_s": t_fm, "evaluations_per_s_fit_multi": float(ne_multi.sum()) / t_fm,
            "evaluations_per_s_lone_fits_1_workspace": ne_lone1 / t_lone1, "evaluations_per_s_lone_fits_4_workspaces": ne_lone4 / t_lone4,
            "speedup_over_lone_1_workspace": (float(ne_multi.sum()) / t_fm) / (ne_lone1 / t_lone1),
            "speedup_over_lone_4_workspaces": (float(ne_multi.sum()) / t_fm) / (ne_lone4 / t_lone4)}
    except Exception as e:  # noqa: BLE001
        res["config5_expert_tuned_fits_8"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    w = rng.random(k) + 0.5
    w /= w.sum()
    gmx = GaussianMixture(w, rng.random((k, d5)), np.array([np.eye(d5) * 0.3] * k), 0.9)
    mix = GpMixture(experts, gmx, "smooth")
    mix.predict_var(xq[:4000])
    t0 = time.perf_counter()
    vm = mix.predict_var(xq)
    t_m = time.perf_counter() - t0
    for e in experts:
        e.close()
    res["config5_mixture_8_experts_1gpu"] = {"predict_var_smooth_points_per_s": m5 / t_m,
                                             "expert_points_per_s": k * m5 / t_m,
                                             "trsm_tflops": k * float(n5) * n5 * m5 / t_m / 1e12,
                                             "checksum": float(vm.sum())}
    # config 2 (n = 4096, d = 8, sq-exp): the serial chain's size -- one evaluation in flight on a one-workspace handle (the
    # reference's `fit` at fixed theta: the whole factorisation is ONE chain launch, kernels_pipe.hip), and twelve in
    # lock-step on a twelve-workspace handle (a round of a tuned fit's COBYLA starts: chain launches per group of panels)
    n2, d2 = 4096, 8
    x2, y2 = workload.make_training_set(n2, d2, seed=42)
    th2 = workload.default_theta(d2) * 3.0
    h = egx.GpHandle(x2, y2, mean=0, corr=0, device=gpu, n_workspaces=1)
    sched_one = h.schedule()
    h.finalize(th2)
    t0 = time.perf_counter()
    for j in range(10):
        h.finalize(th2 * (1.0 + 1e-3 * j))
    t_f2 = (time.perf_counter() - t0) / 10
    tm2 = h.timings()
    h.close()
    h = egx.GpHandle(x2, y2, mean=0, corr=0, device=gpu, n_workspaces=12)
    sched_12 = h.schedule()
    ths2 = np.stack([th2 * (1.0 + 0.01 * c) for c in range(48)])
    h.likelihood_batch(ths2)
    t0 = time.perf_counter()
    h.likelihood_batch(ths2 * 1.001)
    t_b2 = (time.perf_counter() - t0) / 48
    h.close()
    res["config2_sqexp_n4096_d8"] = {
        "fixed_theta_fit_ms_one_in_flight": t_f2 * 1e3, "fits_per_s_one_in_flight": 1.0 / t_f2, "potrf_ms": tm2["potrf_ms"],
        "cholesky_tflops_one_in_flight": tm2["potrf_flops"] / tm2["potrf_ms"] / 1e9,
        "frac_of_fp64_peak_one_in_flight": tm2["potrf_flops"] / tm2["potrf_ms"] / 1e9 / FP64_MFMA_PEAK_TFLOPS,
        "schedule_one_in_flight": sched_one,
        "likelihoods_per_s_lockstep_12": 1.0 / t_b2,
        "frac_of_fp64_peak_lockstep_12": float(n2) ** 3 / 3 / t_b2 / 1e12 / FP64_MFMA_PEAK_TFLOPS,
        "schedule_lockstep_12": sched_12,
        "note": "one in flight: a one-workspace handle, the whole factorisation as one persistent chain launch (16 diagonal blocks of "
                "~62 us + ~15-20 us of device-side hand-offs each; round 4: 57 + 20 + 15..80 us of separate launches per panel); "
                "lock-step 12: a twelve-workspace handle, what a round of a tuned fit's COBYLA starts is"}
    res["size_ladder"] = size_ladder(egx, workload, gpu)
    d6 = 64
    x6, y6 = workload.make_training_set(n, d6, seed=42)
    h = egx.GpHandle(x6, y6, mean=0, corr=0, device=gpu, n_workspaces=1)
    th6 = workload.default_theta(d6)
    h.finalize(th6)
    ts = []
    for j in range(2):
        t0 = time.perf_counter()
        h.finalize(th6 * (1.0 + 0.01 * j))
        ts.append(time.perf_counter() - t0)
    tm = h.timings()
    h.close()
    res["metric_shape_d64_n16384_sqexp"] = {"single_fit_in_flight_fits_per_s": 1.0 / float(np.mean(ts)),
                                            "corr_build_ms": tm["corr_build_ms"],
                                            "corr_build_gbps": tm["corr_bytes"] / tm["corr_build_ms"] / 1e6,
                                            "potrf_ms": tm["potrf_ms"]}
    return res


def moe_sharded_leg(egx, workload, sw, rank, world, gpu):
    """8 experts x n = 8192, d = 16, expert e trained and kept on rank e mod world; predict_var of the smooth mixture on
    100 000 points: every rank evaluates its own experts, ONE all-gather of the partial sums inside the library
    (crates/moe/src/algorithm.rs:167-177 trains the experts serially, :670-685 folds them)."""
    from egobox_amd.moe import GaussianMixture, GpMixture
    k, n5, d5, m5 = 8, 8192, 16, 100000
    rng = np.random.default_rng(5)
    experts = []
    t0 = time.perf_counter()
    for e in range(k):
        if e % world == rank:
            xe, ye = workload.make_training_set(n5, d5, seed=7 + e)
            experts.append(egx.GaussianProcess.params(egx.ConstantMean(), egx.SquaredExponentialCorr())
                           .theta_tuning(egx.ThetaTuning.Fixed(workload.default_theta(d5))).fit(xe, ye))
        else:
            experts.append(None)
    t_fit = time.perf_counter() - t0
    w = rng.random(k) + 0.5
    w /= w.sum()
    gmx = GaussianMixture(w, rng.random((k, d5)), np.array([np.eye(d5) * 0.3] * k), 0.9)
    xq = np.random.default_rng(7).random((m5, d5))
    mix = GpMixture(experts, gmx, "smooth", rank=rank, world=world, sweep=sw)
    mix.predict_var(xq[:4000])
    t0 = time.perf_counter()
    vm = mix.predict_var(xq)
    t_m = time.perf_counter() - t0
    for e in experts:
        if e is not None:
            e.close()
    return {"experts": k, "experts_on_rank_0": sum(1 for e in range(k) if e % world == 0), "fit_own_experts_s": t_fit,
            "predict_var_smooth_points_per_s": m5 / t_m, "expert_points_per_s": k * m5 / t_m,
            "trsm_tflops_all_gpus": k * float(n5) * n5 * m5 / t_m / 1e12, "checksum": float(vm.sum())}


def tuned_fit_leg(egx, sw, d, world):
    """A TUNED fit of the metric's training set through egx_sweep_fit: the reference's 11 multistart COBYLA runs
    (GP_OPTIM_N_START + 1, crates/gp/src/algorithm.rs:928-945) with start s on rank s mod world, bounded at the reference's
    MINIMUM of 25 evaluations per start (GP_COBYLA_MIN_EVAL; its default at d = 32 is 320) so that the leg takes seconds.
    Collective; the fitted model is the same bits for every N."""
    starts = egx.theta_sweep_candidates(11, d, seed=9)  # row 0 = the default theta0 = 0.1, rows 1.. LHS in log10 bounds
    t0 = time.perf_counter()
    evals = sw.fit(starts, [1e-2], [1e1], max_eval=25)
    dt = time.perf_counter() - t0
    m = sw.model()
    lk, s2 = m.fitted_scalars()
    th = m.inner()["theta"]
    return {"starts": 11, "max_eval_per_start": 25, "evaluations": int(evals), "seconds": dt,
            "evaluations_per_s_all_gpus": evals / dt, "n_gpus": world, "likelihood": lk, "sigma2": s2,
            "theta_checksum": float(np.sum(np.log10(th)))}


def kernel_alone_leg(args):
    """Child process of the roofline leg: the same chip-filling launches of one fit, timed the same way, but with the
    look-ahead switched off (EGX_LOOK_MIN beyond n) so that no chain kernel of the next group shares the GPU with them."""
    import egobox_amd as egx
    from egobox_amd import workload
    x, y = workload.make_training_set(args.n, args.d, seed=42)
    base = workload.default_theta(args.d)
    gp = egx.GpHandle(x, y, mean=0, corr=0, device=int(os.environ.get("EGX_BENCH_DEVICE", "0")), n_workspaces=1)
    tims = []
    for j in range(4):
        gp.finalize(base * (1.0 + 0.01 * j))
        tims.append(gp.timings())
    tims = tims[1:]
    ms = float(np.mean([t["potrf_syrk_ms"] for t in tims]))
    fl = float(np.mean([t["syrk_flops"] for t in tims]))
    nl = int(tims[0]["syrk_launches"])
    print(json.dumps({"tflops": fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "launch_ms_avg": ms / max(1, nl), "launches_per_fit": nl,
                      "potrf_ms": float(np.mean([t["potrf_ms"] for t in tims]))}), flush=True)
    gp.close()
    return 0


def run_kernel_alone_leg(args, gpu):
    import subprocess
    env = dict(os.environ, EGX_LOOK_MIN="1000000000", EGX_BENCH_DEVICE=str(gpu))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--kernel-alone-leg", "--npoints", str(args.n),
                              "--dim", str(args.d)], env=env, capture_output=True, text=True, timeout=300)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        r = json.loads(line)
    except Exception as e:  # noqa: BLE001 - a side figure
        return {"error": f"{type(e).__name__}: {e}"[:200]}
    return {"bound": "mfma", "achieved": r["tflops"], "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": r["tflops"] / FP64_MFMA_PEAK_TFLOPS, "launch_ms_avg": r["launch_ms_avg"],
            "launches_per_fit": r["launches_per_fit"], "potrf_ms_with_lookahead_off": r["potrf_ms"],
            "how": "the launches of `roofline_single_matrix` (a lone right-looking fit), timed the same way in a child process with "
                   "EGX_LOOK_MIN beyond n: no look-ahead, so the chain kernels of the next group do not share the GPU with the "
                   "trailing update, which then covers the next group's columns too (and the fit as a whole is slower: "
                   "potrf_ms_with_lookahead_off).  What the kernel does with the chip to itself"}


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run on
    127.0.0.1 (one process per GPU, algorithm.rs:928-945's rayon workers at node scale) and pass its exit code on."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_launch(args, rank, world):
    """CPU rehearsal of the launch path (tests/test_bench_launch_cpu.py): the same spawn / rendezvous / sharding /
    max-over-ranks timing as the real run, gloo instead of RCCL and a STUB evaluator instead of the GPU library.
    The line it prints is marked `dry_launch` and carries no performance number."""
    import torch
    import torch.distributed as dist
    from egobox_amd import sweep as sweep_mod
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    nb = max(1, args.sweep_batch)
    rng = np.random.default_rng(1234)
    cands = rng.uniform(0.05, 0.2, size=((args.steps + args.warmup) * nb, args.d))

    def stub(th):  # stands in for GpHandle.likelihood_batch
        return -np.sum(th * th, axis=1), np.zeros(th.shape[0], dtype=np.int32)
    out_lk = np.zeros(cands.shape[0])
    for i in range(args.warmup):
        sweep_mod.sweep_likelihood(stub, cands[i * nb:(i + 1) * nb])
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        out_lk[i * nb:(i + 1) * nb], _ = sweep_mod.sweep_likelihood(stub, cands[i * nb:(i + 1) * nb])
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    times = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        times = [float(p.item()) for p in parts]
    ok = bool(np.allclose(out_lk[args.warmup * nb:], -np.sum(cands[args.warmup * nb:] ** 2, axis=1)))
    if rank == 0:
        print(json.dumps({"metric": "dry_launch", "value": None, "unit": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_launch": True, "backend": "gloo" if world > 1 else "none",
                          "rank_seconds": times, "results_complete_on_rank0": ok, "fits_per_step": nb}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


# ---- the printed line: numbers first, <= 7 KB (the driver keeps the last 8 KB of stdout) -------------------------------------
DETAILS_FILE = os.path.join("profiles", "bench_last_run_details.json")
_DROP_KEYS = {"metric_note", "thread_settings_tried", "settings_tried", "traffic_source", "how", "kernel", "launch_shape", "note",
              "note_numa", "traffic_measured", "query", "first_handle_in_process", "rccl_version", "pool", "traffic_stale",
              "extra_rows_note", "flops_per_launch_avg", "likelihood_checksum_ok_rows", "correction", "what", "reading", "threads",
              "wall_start", "wall_end"}
_DROP_ORDER = ["cpu_baseline_reference_shaped", "roofline_kernel_alone", "corr_build_roofline", "cpu_baseline_concurrent",
               "pcie_inclusive", "last_step_balance", "stage_ms_single_fit", "vendor_yardstick", "roofline_single_matrix"]


def _schedule_word(s):
    """a handle's schedule (GpHandle.schedule()) as one word"""
    if s.get("flow"):
        form = "flow" if s["flow"] == 1 else "separate+flow-tail"
    elif s.get("whole_factorisation_launch"):
        form = "whole"
    elif s.get("pipelined_chain"):
        form = "chain/group"
    else:
        form = "left" if s.get("left_looking") else "separate"
    return f"{form},g{s.get('panels_per_group')},w{s.get('lockstep')}"


def _compact(o, key=None):
    if isinstance(o, dict):
        if key is not None and key.startswith("schedule") and "panels_per_group" in o:
            return _schedule_word(o)
        return {k: _compact(v, k) for k, v in o.items() if k not in _DROP_KEYS}
    if isinstance(o, (list, tuple)):
        return [_compact(v) for v in o]
    if isinstance(o, float):
        if key is not None and ("checksum" in key or "likelihood" in key):
            return o  # (what the tests and the N > 1 consistency checks compare: every digit)
        return float(f"{o:.5g}") if math.isfinite(o) else None
    if isinstance(o, str) and len(o) > 100 and key not in ("workload", "sample", "error", "collective"):
        return o[:97] + "..."
    return o


def compact_line(out, limit=7000):
    """The full record goes to profiles/bench_last_run_details.json (and gpurun_out/ when it exists); the printed line keeps
    the contract's keys, `roofline`, `cpu_baseline` and the numbers of every leg, without the prose and the settings tables."""
    for path in (os.path.join(ROOT, DETAILS_FILE), os.path.join(ROOT, "gpurun_out", "bench_last_run_details.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    json.dump(out, f, indent=1)
        except OSError:
            pass
    line = _compact(out)
    line["config"]["workload"] = (f"dense GP fixed-theta fit (corr build + Cholesky + likelihood), sq-exp, n={out['config']['n']} "
                                  f"d={out['config']['d']}, LHS + Griewank; theta sweep, {out['config']['sweep_batch_per_step']} per step")
    cb = line.get("cpu_baseline")
    if isinstance(cb, dict) and isinstance(cb.get("sample"), str):
        cb["sample"] = cb["sample"][:160]
    line["details"] = DETAILS_FILE
    for k in _DROP_ORDER:
        if len(json.dumps(line, separators=(",", ":"))) <= limit:
            break
        line.pop(k, None)
    return json.dumps(line, separators=(",", ":"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--npoints", dest="n", type=int, default=16384)
    ap.add_argument("--dim", dest="d", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--in-flight", type=int, default=16,
                    help="candidates in flight per GPU (correlation-matrix workspaces of the sweep handle, 2 GiB each at "
                         "n = 16384): 2 lock-step groups of 8, factored left-looking (round 4; in flight / width -> fits/s: "
                         "8 / 8 40.7, 16 / 8 42.2, 24 / 8 41.9, 32 / 16 42.2; profiles/r04_run6_*.  Round 3, right-looking: "
                         "3 / 1 36.0, 12 / 4 40.7, 24 / 8 41.3, 36 / 12 41.2)")
    ap.add_argument("--sweep-batch", type=int, default=192,
                    help="candidate thetas per step, summed over ALL GPUs (fixed as N grows: strong scaling; 192 = 24 per GPU "
                         "at N = 8, i.e. three lock-step groups of eight per GPU and step, two of them in flight; BASELINE config 4 "
                         "sweeps 512)")
    ap.add_argument("--kernel-alone-leg", action="store_true",
                    help="internal: the roofline launches of one fit with the look-ahead OFF (run as a child process with "
                         "EGX_LOOK_MIN set, the library reads its knobs once); prints one JSON object")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the bounded side measurements of BASELINE configs 3 / 5 and d = 64 (other_configs in the line)")
    ap.add_argument("--lockstep", type=int, default=0,
                    help="candidates factored in lock-step by one launch sequence (0 = library default: 8 from 16 in flight on, else 4)")
    ap.add_argument("--assignment", choices=("static", "dynamic"), default="static",
                    help="candidates -> ranks: c mod N, or pulled from the node-wide counter (egx_sweep_set_assignment)")
    ap.add_argument("--collective", choices=("library", "torch"), default="library",
                    help="who gathers the ranks' results at N > 1: the library's own RCCL all-gather inside "
                         "egx_sweep_likelihood (default), or torch.distributed's (also RCCL) around egx_gp_likelihood_batch. "
                         "The library path is probed once before the warm-up; if it fails on any rank, all ranks switch to "
                         "torch and the line says so (`collective`)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="CPU rehearsal of the launch path: gloo + a stub evaluator, no GPU, no performance number")
    args = ap.parse_args()

    if args.kernel_alone_leg:
        sys.exit(kernel_alone_leg(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); refusing to print a "
                         f"line that would be mislabelled\n")
        sys.exit(2)
    if args.dry_launch:
        sys.exit(dry_launch(args, rank, world))

    os.environ.setdefault("EGX_SWEEP_TIMEOUT_S", "120")  # a benchmark: a peer that is 2 minutes late is gone
    import torch
    import torch.distributed as dist

    if torch.cuda.device_count() < 1:
        sys.stderr.write("bench.py: no GPU visible (the product has no CPU path; --dry-launch rehearses the launch only)\n")
        sys.exit(3)
    gpu = local_rank % torch.cuda.device_count()
    # REHEARSAL of the N > 1 path on a box with fewer GPUs than ranks (EGX_SWEEP_TRANSPORT=shm: the library's all-gather goes
    # through the ranks' shared-memory segment, several ranks share a GPU, torch.distributed runs on gloo).  The line is
    # marked and its throughput means nothing; it exists so that every statement of the N > 1 path has executed once.
    rehearsal = os.environ.get("EGX_SWEEP_TRANSPORT", "") == "shm"
    tdev = "cpu" if rehearsal else f"cuda:{gpu}"
    torch.cuda.set_device(gpu)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")  # only carries the 128 id bytes and the per-rank times
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{gpu}"))

    import egobox_amd as egx
    from egobox_amd import workload

    n, d = args.n, args.d
    x, y = workload.make_training_set(n, d, seed=42)
    # candidates of a theta sweep around the nominal theta = 0.5/sqrt(d): every R is dense and well conditioned.
    # (BASELINE config 4's log-uniform rows in [1e-2, 1e1]^32 mostly give R ~ I: near-zero operands draw less power,
    # the chip clocks up, and the number would flatter the kernel -- they are parity-tested, not timed.)
    base = workload.default_theta(d)
    rng = np.random.default_rng(1234)
    nb = max(1, args.sweep_batch)
    total = (args.steps + args.warmup) * nb
    cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(total, d))

    def any_rank(err):  # the first rank's error string when one has any (the same answer on every rank), else None
        if world == 1:
            return err
        box = [None] * world
        dist.all_gather_object(box, err)
        for r, e in enumerate(box):
            if e is not None:
                return f"rank {r}: {e}"
        return None

    # uploads happen here: inputs resident.  Rank 0 draws the RCCL unique id; torch's store carries its 128 bytes
    use_lib = args.collective == "library" or world == 1
    collective = "library: ncclAllGather inside egx_sweep_likelihood" if use_lib else "torch.distributed all_gather (--collective torch)"
    sw, create_err = None, None
    if use_lib:
        try:
            sw = egx.rendezvous_sweep(x, y, device=gpu, n_workspaces=max(1, args.in_flight))
        except Exception as e:  # noqa: BLE001 - at N > 1 the ranks agree on the torch path below
            if world == 1:
                raise
            create_err = f"{type(e).__name__}: {e}"[:200]
        create_err = any_rank(create_err)
        if create_err is not None:
            use_lib = False
            collective = f"torch.distributed all_gather (egx_sweep_create failed on a rank: {create_err})"
            if sw is not None:
                sw.close()
                sw = None
    if sw is None:  # this rank's replica without a communicator of the library's own
        sw = egx.Sweep(x, y, device=gpu, rank=0, world=1, id_bytes=None, n_workspaces=max(1, args.in_flight))
    lockstep = sw.set_lockstep(args.lockstep)
    if args.assignment == "dynamic" and use_lib:
        sw.set_assignment(True)
    lkhs = np.zeros(total)
    stats = np.zeros(total, dtype=np.int32)
    torch_eval_s = [0.0]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def torch_evaluate(th):
        te0 = time.perf_counter()
        r = sw.local_likelihood_batch(th)
        torch_eval_s[0] = time.perf_counter() - te0
        return r

    def step(i):
        sl = slice(i * nb, (i + 1) * nb)
        if use_lib:
            lkhs[sl], stats[sl] = sw.likelihood(cands[sl])  # collective: shard k mod N + ncclAllGather inside the library
        else:  # the same shard (k mod N) evaluated by egx_gp_likelihood_batch, gathered by torch's RCCL
            lkhs[sl], stats[sl] = egx.sweep_likelihood(torch_evaluate, cands[sl], rank, world,
                                                             device=None if rehearsal else tdev)

    if world > 1 and use_lib:
        # PROBE, untimed: one candidate per rank through the library's collective.  The world > 1 RCCL path of the library
        # cannot be exercised on the one-GPU boxes it is developed on; if it fails HERE, on any rank, every rank switches to
        # torch's all-gather so that the run still yields its line -- and the line names what happened.
        probe_err = None
        try:
            sw.likelihood(cands[:world])
            if os.environ.get("EGX_BENCH_FAIL_PROBE") == str(rank):  # test hook (tests/test_gpu_configs.py)
                raise RuntimeError("injected probe failure")
        except Exception as e:  # noqa: BLE001
            probe_err = f"{type(e).__name__}: {e}"[:200]
        probe_err = any_rank(probe_err)
        if probe_err is not None:
            use_lib = False
            collective = f"torch.distributed all_gather (the library's collective failed its probe on a rank: {probe_err})"

    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    rank_seconds = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        rank_seconds = [float(p.item()) for p in parts]
        elapsed = max(rank_seconds)  # MAX over ranks
    if use_lib:
        per_rank_last, eval_s_last = sw.last_balance()
    else:
        per_rank_last, eval_s_last = [len(range(r, nb, world)) for r in range(world)], torch_eval_s[0]
    eval_seconds = [eval_s_last]
    if world > 1:
        t = torch.tensor([eval_s_last], dtype=torch.float64, device=tdev)
        parts = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        eval_seconds = [float(p.item()) for p in parts]
    info = sw.info()
    # BASELINE config 5 sharded over the ranks (one expert per GPU at N = 8): a bounded side leg AFTER the timed region, on
    # every rank (it is a collective): expert e on rank e mod N, smooth recombination of predict_var on 100 000 points
    # through egx_moe_predict_valvar and the sweep's communicator.  Never part of `value`; any failure becomes a string.
    moe_sharded = None
    if world > 1 and not args.no_extra_configs and use_lib:
        try:
            moe_sharded = moe_sharded_leg(egx, workload, sw, rank, world, gpu)
        except Exception as e:  # noqa: BLE001 - reported in the line, must not take the headline number down
            moe_sharded = {"error": f"{type(e).__name__}: {e}"[:300]}
    # the tuned fit with its starts sharded over the ranks (every N, also 1): bounded, after the timed region, collective
    tuned = None
    if not args.no_extra_configs and use_lib:
        try:
            tuned = tuned_fit_leg(egx, sw, d, world)
        except Exception as e:  # noqa: BLE001
            tuned = {"error": f"{type(e).__name__}: {e}"[:300]}
    # BASELINE config 4 in full: all 512 rows of the log-uniform sweep theta_sweep_candidates(512, d) (optimization.rs:49-66)
    # through the same collective, after the timed region (most rows give R ~ I and a few are not positive definite: it is
    # a parity / coverage figure -- tests/test_gpu_configs.py pins 33 of its rows -- never `value`)
    config4 = None
    if not args.no_extra_configs and (n, d) == (16384, 32):
        try:
            c4 = egx.theta_sweep_candidates(512, d)
            barrier()
            t4 = time.perf_counter()
            if use_lib:
                l4, s4 = sw.likelihood(c4)
            else:
                l4, s4 = egx.sweep_likelihood(torch_evaluate, c4, rank, world, device=None if rehearsal else tdev)
            barrier()
            t4 = time.perf_counter() - t4
            okc = s4 == 0
            config4 = {"rows": 512, "seconds": t4, "evals_per_s_all_gpus": 512 / t4, "n_gpus": world,
                       "status_counts": {str(int(v)): int(np.sum(s4 == v)) for v in np.unique(s4)},
                       "likelihood_checksum_ok_rows": float(np.sum(l4[okc])),
                       "best_row": int(np.argmax(np.where(okc, l4, -np.inf))),
                       "best_likelihood": float(np.max(np.where(okc, l4, -np.inf)))}
        except Exception as e:  # noqa: BLE001
            config4 = {"error": f"{type(e).__name__}: {e}"[:300]}
    # What ONE rank sees at N = 8 (no 8-GPU node is available to this build; the driver's SCALE run is the measurement):
    # 192 per step over 8 GPUs = 24 candidates per rank and step = three lock-step groups of eight, two in flight, then one --
    # and the all-gather after every step.  This rank drives exactly that through the same collective entry point; the implied
    # 8-GPU figure is 8 x its rate (candidates are independent, X and y replicated: nothing else is shared between ranks).
    n8_emulation = None
    if world == 1 and use_lib and not args.no_extra_configs and nb >= 24:
        try:
            share = nb // 8
            c8 = cands[:12 * share].reshape(12, share, d)
            for i in range(2):
                sw.likelihood(c8[i])
            barrier()
            t8 = time.perf_counter()
            for i in range(2, 12):
                sw.likelihood(c8[i])
            barrier()
            t8 = (time.perf_counter() - t8) / 10
            n8_emulation = {"candidates_per_rank_and_step": share, "steps": 10, "ms_per_step": t8 * 1e3,
                            "fits_per_s_this_rank": share / t8, "implied_8gpu_fits_per_s": 8 * share / t8,
                            "implied_scaling_over_n1_value": 8 * share / t8 / (args.steps * nb / elapsed),
                            "drain_cost_frac": 1.0 - share / t8 / (args.steps * nb / elapsed),
                            "what": "one rank's share of a 192-candidate step at N = 8 (three groups of eight: two in flight, then "
                                    "one), an all-gather per step; 8 x this rank's rate"}
        except Exception as e:  # noqa: BLE001
            n8_emulation = {"error": f"{type(e).__name__}: {e}"[:300]}
    sw.close()
    if use_lib and info["rccl_ranks"] != world and not rehearsal:
        sys.stderr.write(f"bench.py: the library's RCCL communicator has {info['rccl_ranks']} ranks, world is {world}\n")
        sys.exit(4)

    if rank == 0:
        # ---- roofline leg (not part of `value`): ONE fit in flight on its own handle, so the HIP-event duration of
        # every update-kernel launch on its stream is not overlapped by another candidate's kernels
        gp = egx.GpHandle(x, y, mean=0, corr=0, device=gpu, n_workspaces=1)
        t_fit, tim0 = [], []
        for j in range(4):  # the product's schedule: what a lone fit costs
            tf0 = time.perf_counter()
            gp.finalize(base * (1.0 + 0.01 * j))
            t_fit.append(time.perf_counter() - tf0)
            tim0.append(gp.timings())
        t_fit, tim0 = t_fit[1:], tim0[1:]
        # ... and the same fits with the look-ahead columns' update back IN FRONT of the trailing update on the main stream
        # (in the product it runs beside it on a side stream and overlaps the HIP events around it): clean per-launch durations
        prev_side = egx.set_tuning("lur_side", 0)
        tim1 = []
        for j in range(3):
            gp.finalize(base * (1.0 + 0.01 * (4 + j)))
            tim1.append(gp.timings())
        egx.set_tuning("lur_side", prev_side)
        fits = args.steps * nb
        potrf_ms = float(np.mean([t["potrf_ms"] for t in tim0]))
        corr_ms = float(np.mean([t["corr_build_ms"] for t in tim0]))
        solve_ms = float(np.mean([t["solve_ms"] for t in tim0]))
        host_ms = float(np.mean([t["host_ms"] for t in tim0]))
        flops = tim1[0]["potrf_flops"]
        tflops = flops / (potrf_ms * 1e-3) / 1e12
        syrk_ms = float(np.mean([t["potrf_syrk_ms"] for t in tim1]))
        syrk_flops = float(np.mean([t["syrk_flops"] for t in tim1]))
        syrk_launches = int(tim1[0]["syrk_launches"])
        syrk_tflops = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        # ---- ONE lock-step group alone (the unit the timed region keeps two of in flight): its whole factorisation rate.
        # (No per-launch figure for the batched launches: inside a group the rest-of-group update runs on a side stream and
        # overlaps the HIP events around the trailing update -- 47.7 TFLOP/s "per launch" beside 58.8 for the whole group.)
        gl = max(1, min(lockstep, 8))
        grp, roof_group = None, None
        if gl > 1:
            gq = egx.GpHandle(x, y, mean=0, corr=0, device=gpu, n_workspaces=gl)
            gq.set_lockstep(gl)

            def group_batch(j):
                gq.likelihood_batch(np.stack([base * (1.0 + 0.01 * (j * gl + c)) for c in range(gl)]))
                return gq.timings()
            timg = [group_batch(j) for j in range(4)]
            g_potrf = float(np.mean([t["potrf_ms"] for t in timg[1:]]))
            grp = {"matrices": gl, "potrf_ms_all_matrices": g_potrf, "cholesky_tflops": gl * flops / (g_potrf * 1e-3) / 1e12,
                   "frac_of_fp64_peak": gl * flops / (g_potrf * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS}
            # ---- the launch the timed region issues: the dominant update of a lock-step group (grid.z = gl matrices).  A
            # handle of this shape factors LEFT-looking over its panel groups (round 4): the long update of the next group's
            # columns is alone on its stream, HIP events bracket it cleanly, the previous group's chain shares the chip as in
            # the product.  (A right-looking handle -- EGX_POTRF_LEFT=0 -- runs the look-ahead columns' update beside the
            # trailing update on a side stream; lur_side = 0 puts it back in front for this leg.)
            try:
                prev = egx.set_tuning("lur_side", 0)  # (right-looking handles only: the left-looking schedule has no LUr)
                timq = [group_batch(10 + j) for j in range(3)][1:]
                egx.set_tuning("lur_side", prev)
                ms8 = float(np.mean([t["potrf_syrk_ms"] for t in timq]))
                fl8 = float(np.mean([t["syrk_flops"] for t in timq]))
                nl8 = int(timq[0]["syrk_launches"])
                if ms8 > 0 and nl8 > 0:
                    roof_group = {"tflops": fl8 / (ms8 * 1e-3) / 1e12, "launches": nl8, "launch_ms_avg": ms8 / nl8,
                                  "flops_per_launch_avg": fl8 / nl8, "share_of_potrf_flops": fl8 / (gl * flops),
                                  "potrf_ms_all_matrices_serialised": float(np.mean([t["potrf_ms"] for t in timq]))}
            except Exception as e:  # noqa: BLE001 - falls back to the single-matrix figure below
                sys.stderr.write(f"bench.py: group roofline leg failed: {e}\n")
            gq.close()
        traffic, pmc = measured_traffic(n, d)
        # the same counters for the launch `roofline` describes (8 matrices, left-looking long update): committed summary
        # (an OFFLINE measurement: attached only while it still describes this build -- the kernel's source text has the hash
        #  the summary was taken with and the live launch shape is the measured one; otherwise `traffic` is null)
        ll_pmc, ll_stale = None, None
        import glob
        ll_found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_lockstep_group_left_looking_summary.json")))
        ll_path = ll_found[-1] if ll_found else ""  # the latest round's measurement
        if (n, d) == (16384, 32) and ll_path:
            with open(ll_path) as f:
                ll_pmc = json.load(f)
            sig = stream_kernel_signature()
            if sig != ll_pmc.get("kernel_source_sha256"):
                ll_stale = f"k_gemm_stream source changed since the counters were taken ({sig[:12]} != recorded)"
            elif roof_group is not None and (roof_group["launches"] != ll_pmc.get("launches_per_factorisation") or
                                             abs(roof_group["flops_per_launch_avg"] / ll_pmc["algorithmic_flops_per_launch"] - 1.0) > 1e-6):
                ll_stale = "the live launch shape (launches per factorisation / flops per launch) is not the measured one"
            if ll_stale:
                sys.stderr.write(f"bench.py: roofline.traffic not attached: {ll_stale}\n")
                ll_pmc = None
        ok = stats[args.warmup * nb:] == 0
        out = {
            "metric": "gp_fixed_theta_fits_per_sec", "value": fits / elapsed, "unit": "fits/s",
            "metric_note": "one fit = correlation build + Cholesky with fused forward solves + GLS + reduced likelihood = one "
                           "evaluation of the objective the reference's optimiser multiplies (algorithm.rs:880-897, 988-1056), "
                           "timed through egx_sweep_likelihood; the gamma back-substitution of algorithm.rs:1034 (1 ms, once "
                           "per tuned fit, for the winner) is in stage_ms_single_fit.gamma_solve and in "
                           "single_fit_in_flight_fits_per_s / pcie_inclusive, which time egx_gp_finalize; the CPU baselines "
                           "include it (n^2 flops beside n^3/3)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"dense GP fixed-theta fit (correlation build + Cholesky + likelihood), squared "
                                   f"exponential, n={n} d={d}, classic LHS + Griewank (BASELINE metric line); a "
                                   f"theta sweep around 0.5/sqrt(d), {nb} candidates per step over all GPUs",
                       "n": n, "d": d, "corr": "SquaredExponential", "mean": "Constant",
                       "parallelism": f"sweep-dp{world}", "sweep_batch_per_step": nb,
                       "fits_in_flight_per_gpu": max(1, args.in_flight), "lockstep_width": lockstep},
            "fits_per_step": nb,
            "rccl_ranks": info["rccl_ranks"], "rccl_version": info["rccl_version"], "collective": collective,
            **({"rehearsal": "EGX_SWEEP_TRANSPORT=shm: host transport instead of RCCL, ranks share GPUs -- NOT a measurement"}
               if rehearsal else {}),
            "allgathers_in_timed_region": args.steps, "assignment": args.assignment,
            "rank_seconds": rank_seconds,
            "last_step_balance": {"candidates_per_rank": [int(v) for v in per_rank_last],
                                  "evaluation_seconds_per_rank": eval_seconds,
                                  "imbalance_max_over_mean": (max(eval_seconds) / (sum(eval_seconds) / len(eval_seconds))
                                                              if sum(eval_seconds) > 0 else None)},
            "cholesky_tflops_per_gpu_in_timed_region": fits * flops / elapsed / 1e12 / world,
            "stage_ms_single_fit": {"corr_build": corr_ms, "potrf_fused_fwd_solve": potrf_ms, "gamma_solve": solve_ms,
                                    "host_gls": host_ms},
            "single_fit_in_flight_fits_per_s": 1.0 / float(np.mean(t_fit)),
            "cholesky_tflops_single_fit": tflops,
            "roofline": (None if roof_group is None else {
                "bound": "mfma", "achieved": roof_group["tflops"], "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": roof_group["tflops"] / FP64_MFMA_PEAK_TFLOPS,
                "traffic": (ll_pmc["corrected_traffic_bytes_per_launch"] if ll_pmc and gl == 8 else None),
                "traffic_algorithmic_bytes": (ll_pmc["algorithmic_bytes_per_launch"] if ll_pmc and gl == 8 else None),
                "launch_shape": f"{gl} matrices per launch (grid.z): the launch the timed region issues for a lock-step group",
                "kernel": "k_gemm_stream<LOWER, TAG 1>: the LONG left-looking update of a lock-step group's next 1024 columns "
                          "(C -= L[:, 0:g) L[cols, 0:g)^T with K = every column before the previous group, 128x256 tiles; a "
                          "handle with n >= 14336 and a lock-step width of 8 factors left-looking over its groups of four "
                          "256-wide panels, kernels_chol.hip launch_potrf): "
                          f"{100.0 * roof_group['share_of_potrf_flops']:.0f} % of the factorisations' n^3/3 flops",
                "share_of_potrf_flops": roof_group["share_of_potrf_flops"],
                "launches_per_group": roof_group["launches"], "launch_ms_avg": roof_group["launch_ms_avg"],
                "flops_per_launch_avg": roof_group["flops_per_launch_avg"],
                "how": "algorithmic flops (matrices * 2 * K * (rows * 1024 - 1024 * 1023 / 2) per launch) / HIP-event durations "
                       "around every such launch on the stream it is launched on; ONE lock-step group in flight; the chain of "
                       "the previous group (diagonal blocks, panel solves, in-group updates) shares the chip with it as in "
                       "the product.  Reproduce: rocprofv3 --kernel-trace --stats -- python tools/group_roofline.py, row "
                       "'left-looking long update' of tools/rocpd_stats.py (profiles/r04_group_roofline_kernel_stats.txt)",
                "traffic_stale": ll_stale,
                "traffic_source": (None if not ll_pmc else {
                    "measured": "OFFLINE, not in this run; attached because the kernel source hash and the live launch shape match "
                                "the measurement",
                    "kernel_source_sha256": ll_pmc.get("kernel_source_sha256"),
                    "file": "profiles/" + os.path.basename(ll_path) + " (from ..._pass1-3.json: separate rocprofv3 --pmc passes of "
                            "tools/group_roofline.py, tools/gpu_pmc.sh group + tools/pmc_group_summary.py, offline)",
                    "fetch_size_bytes_per_launch_raw": ll_pmc["fetch_bytes_per_launch"],
                    "write_size_bytes_per_launch_raw": ll_pmc["write_bytes_per_launch"],
                    "correction": ll_pmc["correction"], "l2_hit_rate": ll_pmc["l2_hit_rate"],
                    "mfma_busy_frac": ll_pmc["mfma_busy_frac"],
                    "note": "3.8x the algorithmic bytes: each 128-row block of the factor is fetched by the four column tiles "
                            "that use it on four different XCDs; 13.9 GB in 11 ms = 1.3 TB/s of fabric traffic (mostly "
                            "Infinity-Cache hits), far from the bound -- and both walks that share those rows inside one XCD's "
                            "L2 were slower (profiles/r04_run2_stream_walk_*, r04_run18_*)"})}),
            "roofline_single_matrix": {"bound": "mfma", "achieved": syrk_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": syrk_tflops / FP64_MFMA_PEAK_TFLOPS,
                         "traffic": traffic,
                         "launch_shape": "ONE matrix per launch (a lone right-looking fit; the definition of rounds 1 to 3), with the "
                                         "look-ahead columns' update serialised in front of it for this leg (lur_side = 0; the "
                                         "product runs it beside the trailing update).  The timed region launches left-looking "
                                         "updates for lock-step groups of eight: `roofline`",
                         "kernel": "k_gemm_stream<LOWER> (Cholesky trailing update C -= P P^T, 128x256 tiles, once per group "
                                   "of four 256-wide panels, K = 1024, at this size; the launches with >= 512 tiles: "
                                   f"{100.0 * syrk_flops / flops:.0f} % of the factorisation's n^3/3 flops)",
                         "share_of_potrf_flops": syrk_flops / flops,
                         "launches_per_fit": syrk_launches, "launch_ms_avg": syrk_ms / max(1, syrk_launches),
                         "flops_per_launch_avg": syrk_flops / max(1, syrk_launches),
                         "how": "algorithmic flops (2*K*ncols*(ncols+1)/2 per launch, K = group width) / HIP-event durations around every "
                                "launch on the stream it is launched on, one fit in flight (separate leg after the timed "
                                "region; in the timed region the candidates in flight overlap and share the GPU)",
                         "traffic_measured": "offline: separate rocprofv3 --pmc passes (tools/gpu_pmc.sh lone), committed under "
                                             "profiles/ and READ here, not measured by this run",
                         "traffic_source": (None if pmc is None else
                                            {"file": os.path.relpath(PMC_SUMMARY, ROOT),
                                             "fetch_bytes_per_launch_raw": pmc["fetch_bytes_per_launch"],
                                             "write_bytes_per_launch_raw": pmc["write_bytes_per_launch"],
                                             "c_read_bytes_per_launch_algorithmic": pmc["c_read_bytes_per_launch"],
                                             "correction": "gfx950 FETCH_SIZE counts 16 B/lane reads at half: traffic = "
                                                           "WRITE + C_read + 2 * (FETCH - C_read)"})},
            "lockstep_group_alone": grp,
            "roofline_kernel_alone": (run_kernel_alone_leg(args, gpu) if world == 1 and not args.no_extra_configs else None),
            "vendor_yardstick": (vendor_yardstick(torch, gpu) if world == 1 and not args.no_extra_configs else None),
            "corr_build_gbps": tim1[0]["corr_bytes"] / (corr_ms * 1e-3) / 1e9,
            "corr_build_roofline": {"bound": "hbm", "achieved": tim1[0]["corr_bytes"] / (corr_ms * 1e-3) / 1e9,
                                    "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": tim1[0]["corr_bytes"] / (corr_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "traffic": (None if pmc is None or "k_corr_sym" not in pmc else
                                                pmc["k_corr_sym"].get("write_size_bytes_per_launch_raw", 0.0)
                                                + pmc["k_corr_sym"].get("fetch_size_bytes_per_launch_raw", 0.0)),
                                    "note": "K1 is bound by FP64 VALU issue at d = 32 (3 d + ~25 ops per pair), not by "
                                            "its 1.08 GB of writes: DESIGN.md section 4"},
            "candidates_ok": int(ok.sum()), "candidates_failed": int((~ok).sum()),
            "likelihood_checksum": float(np.sum(lkhs[args.warmup * nb:][ok])),
        }
        if out["roofline"] is None:  # (a lock-step width of one: the single-matrix launches are the timed region's)
            out["roofline"] = out["roofline_single_matrix"]
        gp.close()
        if world == 1:
            # the boundary hands over HOST buffers: one-shot call sequences create -> fit -> read the scalars -> destroy,
            # as the reference's one-shot `fit` is used (a model per expert / per EGO iteration) -- never `value`.
            # `first_handle_in_process`: nothing pooled (2 GiB device allocation, stream / event creation, the first
            # factorisation on untouched memory); `pooled`: the resources a destroyed handle of this shape left behind
            def one_shot():
                tc0 = time.perf_counter()
                hc = egx.GpHandle(x, y, mean=0, corr=0, device=gpu, n_workspaces=1)
                tc1 = time.perf_counter()
                hc.finalize(base)
                hc.fitted_scalars()
                tc2 = time.perf_counter()
                hc.close()
                tc3 = time.perf_counter()
                return {"create_alloc_upload_s": tc1 - tc0, "fit_and_download_s": tc2 - tc1, "destroy_s": tc3 - tc2,
                        "fits_per_s": 1.0 / (tc3 - tc0)}
            egx.trim()
            first = one_shot()
            pooled = [one_shot() for _ in range(4)][1:]
            pooled_med = sorted(pooled, key=lambda r: r["fits_per_s"])[len(pooled) // 2]
            out["pcie_inclusive"] = {"first_handle_in_process": first, "pooled": pooled_med,
                                     "fits_per_s_cold_handle": pooled_med["fits_per_s"],
                                     "pool": egx.pool_stats(),
                                     "note": "x, y (4 MiB) cross PCIe once per handle; every further fit on the handle "
                                             "moves (p + 2) n doubles back (0.4 MB); destroyed handles leave their "
                                             "device resources in the library's pool (egx_trim frees it)"}
        if world == 1 and not args.no_extra_configs:
            out["other_configs"] = other_configs(egx, workload, gpu)
        if moe_sharded is not None:
            out["other_configs"] = {"config5_mixture_8_experts_sharded": moe_sharded}
        if tuned is not None:
            out.setdefault("other_configs", {})["tuned_fit_11_starts_sharded"] = tuned
        if config4 is not None:
            out.setdefault("other_configs", {})["config4_sweep_512"] = config4
        if n8_emulation is not None:
            out["n8_rank_emulation"] = n8_emulation
        if not args.no_cpu_baseline and world == 1:
            torch.cuda.synchronize()
            cb = cpu_baseline(n, d)
            hp = host_fp64_peak()
            if hp is not None and "dpotrf_gflops" in cb:
                cb["host_fp64_peak"] = hp
                cb["dpotrf_frac_of_host_fp64_peak"] = cb["dpotrf_gflops"] / hp["peak_gflops"]
                cb["note_numa"] = ("numactl is not in this image: memory placement is first-touch by the OpenMP correlation "
                                   "build (all host threads, static schedule), i.e. spread over both sockets")
            out["cpu_baseline"] = cb
            if "value" in cb:
                # LATENCY mode against latency mode: one fit in flight on the GPU, one fit on the CPU's best thread count
                out["speedup_latency_mode_single_fit_vs_cpu_baseline"] = out["single_fit_in_flight_fits_per_s"] / cb["value"]
                best_threads = min(cb.get("thread_settings_tried", [{"blas_threads": 16, "fit_s": 0}]),
                                   key=lambda r: r["fit_s"])["blas_threads"]
                cc = cpu_baseline_concurrent(n, d, int(best_threads) if str(best_threads).isdigit() else 16)
                out["cpu_baseline_concurrent"] = cc
                if "value" in cc:
                    # THROUGHPUT mode against throughput mode: `value` (several candidates in flight) over the BEST CPU
                    # throughput measured (simultaneous fits, or one fit after the other when running them side by side is slower)
                    best_cpu = max(cc["value"], cb["value"])
                    out["cpu_best_throughput_fits_per_s"] = best_cpu
                    out["speedup_throughput_mode_vs_best_cpu_throughput"] = out["value"] / best_cpu
                out["speedup_mixed_modes_value_vs_single_cpu_fit"] = out["value"] / cb["value"]
            rs = cpu_baseline_reference_shaped(n, d)
            if rs is not None:
                out["cpu_baseline_reference_shaped"] = rs
        print(compact_line(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
