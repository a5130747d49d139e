#!/usr/bin/env python3
"""Headline benchmark: fixed-theta GP fits per second, n = 16384, d = 32, squared exponential.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one candidate theta on every GPU:
correlation-matrix build (K1) + blocked FP64-MFMA Cholesky with fused forward solves (K3/K4) +
GLS / reduced likelihood + gamma back-substitution = one `ThetaTuning::Fixed` fit
(crates/gp/src/algorithm.rs:869-872, 966-978), the unit the reference's COBYLA multiplies.  The
training set is resident in HBM before the timed region starts.  With N GPUs each rank fits a different
candidate of the theta sweep (weak scaling, no data-path collective) and one RCCL all-gather of the
(likelihood, status) pairs closes every step (egobox_amd/sweep.py).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the Cholesky trailing update,
FP64 MFMA bound); `cpu_baseline` times the numpy/scipy oracle (OpenBLAS, all host cores) on a bounded
sample of the same workload on the GPU box's host.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# public MI355X figures (the CDNA4 guide in this image lists no FP64 matrix peak):
FP64_MFMA_PEAK_TFLOPS = 78.6
HBM_PEAK_GBPS = 8000.0


def cpu_baseline(n_full, d, seconds_budget=30.0):
    """Oracle ('port': numpy restatement, LAPACK dpotrf/dtrtrs through scipy -- the reference's `blas`
    feature shape) timed on a bounded sample, extrapolated to the full size with the measured
    O(n^2 d) / O(n^3) split."""
    from oracle import gp_oracle as O
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    n_s = min(n_full, 8192)  # ~20 s of host work on the GPU box (n^3 / 8 of the full fit)
    x = O.lhs_classic(n_s, d, 42)
    y = O.griewank(x)
    theta = np.full(d, 0.5 / np.sqrt(d))
    _, _, xn, _, _, yn, _, ys, fx = O.prepare_training(x, y)
    t0 = time.perf_counter()
    r_mx = O.corr_matrix_dense(O.SQEXP, xn, theta, np.eye(d), O.DEFAULT_NUGGET)
    t1 = time.perf_counter()
    lk, _ = O.reduced_likelihood_from_r(fx, r_mx, yn, ys[0])
    t2 = time.perf_counter()
    t_corr, t_chol = t1 - t0, t2 - t1
    s = n_full / n_s
    t_full = t_corr * s ** 2 + t_chol * s ** 3
    return {
        "value": 1.0 / t_full, "unit": "fits/s", "cores": int(cores), "kind": "port",
        "sample": (f"oracle (numpy+scipy/OpenBLAS) fixed-theta fit at n={n_s}, d={d}: corr build {t_corr:.3f}s + "
                   f"cholesky/solves {t_chol:.3f}s; extrapolated to n={n_full} with n^2 / n^3 scaling "
                   f"({t_full:.1f}s per fit)"),
        "sample_seconds": t2 - t0, "sample_likelihood": lk,
    }


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
                       f"({ts[0]:.2f}s) and n={ns[1]} ({ts[1]:.2f}s), d={d}; a n^2 + b n^3 extrapolated to n={n_full} "
                       f"({t_full:.0f}s per fit; the table alone would need {n_full * (n_full - 1) // 2 * d * 8 / 1e9:.1f} GB)"),
            "sample_seconds": float(sum(ts)), "sample_status": int(r["status"])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--npoints", dest="n", type=int, default=16384)
    ap.add_argument("--dim", dest="d", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; "
                    "gloo only for checking the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--batch", type=int, default=2,
                    help="candidate thetas in flight per GPU and step (independent fits on separate workspaces/streams, "
                         "like the reference's rayon multistart, crates/gp/src/algorithm.rs:928-945)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    gpu = local_rank % max(1, ndev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(gpu)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{gpu}"))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(0)
    # collective payloads live on the GPU under RCCL, on the host under gloo
    dev = torch.device(f"cuda:{gpu}") if args.backend == "nccl" else None
    local_rank = gpu

    import egobox_amd as egx
    from egobox_amd import workload

    n, d = args.n, args.d
    x, y = workload.make_training_set(n, d, seed=42)
    # candidates of the theta sweep around the nominal theta (a different one every step and rank)
    base = workload.default_theta(d)
    rng = np.random.default_rng(1234)
    nb = max(1, args.batch)
    total = (args.steps + args.warmup) * world * nb
    cands = base * 10.0 ** rng.uniform(-0.15, 0.15, size=(total, d))

    # one handle (own workspace + stream pair) per in-flight candidate; uploads happen here: inputs resident
    gps = [egx.GpHandle(x, y, mean=0, corr=0, device=local_rank, n_workspaces=1) for _ in range(nb)]
    gp = gps[0]
    lkhs = np.zeros(total)
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(nb)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_fit(i, b):
        idx = (i * world + rank) * nb + b
        g = gps[b]
        g.finalize(cands[idx])  # the fixed-theta fit (ctypes releases the GIL: the nb fits overlap on the GPU)
        lkhs[idx] = g.fitted_scalars()[0]
        return g.timings()  # struct copy of the HIP-event stage durations

    def step(i):
        if nb == 1:
            return one_fit(i, 0)
        return list(pool.map(lambda b: one_fit(i, b), range(nb)))[0]

    tim = []
    for i in range(args.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        tim.append(step(i))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        # the one exchange of the sweep: all-gather of the per-candidate results (16 B each), and
        # the max-over-ranks clock
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dev is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        from egobox_amd.sweep import sweep_likelihood
        lk2 = lkhs.reshape(-1, world, nb)  # (step, rank, in-flight slot)
        rows = cands.reshape(-1, world, nb, d)
        for b in range(nb):
            mine = lk2[:, rank, b].copy()
            all_lk, _ = sweep_likelihood(lambda th: (mine, np.zeros(len(mine), dtype=np.int32)),
                                         rows[:, :, b, :].reshape(-1, d), device=dev)
            lk2[:, :, b] = all_lk.reshape(-1, world)

    # ---- roofline leg (not part of `value`): ONE fit in flight, so the HIP-event duration of the factorisation
    # on its stream is not overlapped by another candidate's kernels
    tim1 = []
    for j in range(3):
        gp.finalize(cands[j])
        tim1.append(gp.timings())
    if rank == 0:
        fits = args.steps * world * nb
        potrf_ms = float(np.mean([t["potrf_ms"] for t in tim1]))
        corr_ms = float(np.mean([t["corr_build_ms"] for t in tim1]))
        solve_ms = float(np.mean([t["solve_ms"] for t in tim1]))
        host_ms = float(np.mean([t["host_ms"] for t in tim1]))
        flops = tim1[0]["potrf_flops"]
        tflops = flops / (potrf_ms * 1e-3) / 1e12
        syrk_ms = float(np.mean([t["potrf_syrk_ms"] for t in tim1]))
        syrk_flops = float(np.mean([t["syrk_flops"] for t in tim1]))
        syrk_launches = int(tim1[0]["syrk_launches"])
        syrk_tflops = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        agg_tflops = fits * flops / elapsed / 1e12 / world
        out = {
            "metric": "gp_fixed_theta_fits_per_sec", "value": fits / elapsed, "unit": "fits/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"dense GP fixed-theta fit, squared exponential, n={n} d={d}, classic LHS + "
                                   "Griewank (BASELINE metric line / configs[2] size); --batch candidate thetas in "
                                   "flight per GPU per step",
                       "n": n, "d": d, "corr": "SquaredExponential", "mean": "Constant",
                       "parallelism": f"sweep-dp{world}", "fits_in_flight_per_gpu": nb},
            "fits_per_step": world * nb,
            "cholesky_tflops_per_gpu_in_timed_region": agg_tflops,
            "stage_ms_single_fit": {"corr_build": corr_ms, "potrf_fused_fwd_solve": potrf_ms, "gamma_solve": solve_ms,
                                    "host_gls": host_ms},
            "cholesky_tflops_single_fit": tflops,
            "roofline": {"bound": "mfma", "achieved": syrk_tflops, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": syrk_tflops / FP64_MFMA_PEAK_TFLOPS,
                         # HBM/fabric bytes per launch from the PMC passes of this very command (n = 16384, d = 32 only)
                         "traffic": (1.30e9 if (n, d) == (16384, 32) else None),
                         "kernel": "k_gemm_nt_sub<LOWER,128,256,64,64,512> (Cholesky trailing update C -= P P^T once per "
                                   "group of two 256-wide panels, K = 512; the launches that fill the chip, ~83% of the "
                                   "factorisation's flops)",
                         "launches_per_fit": syrk_launches, "launch_ms_avg": syrk_ms / max(1, syrk_launches),
                         "flops_per_launch_avg": syrk_flops / max(1, syrk_launches),
                         "how": "algorithmic flops (2*K*ncols*(ncols+1)/2 per launch) / HIP-event durations around every "
                                "launch on the stream it is launched on, one fit in flight (separate leg after the timed "
                                "region; in the timed region two candidates overlap and share the GPU)",
                         "traffic_measured_offline": "rocprofv3 --pmc (separate passes, same command): FETCH_SIZE 0.817 GB + "
                                                     "WRITE_SIZE 0.485 GB per launch = 1.30 GB against ~1.0 GB algorithmic "
                                                     "(C tile read + write + panel), L2 hit rate 0.62, MFMA busy 0.71 -- "
                                                     "profiles/r01_run19_pmc_wide_kernel_summary.txt",
                         "measured_mfma_f64_ceiling_tflops": "77.6 register-only, 77.3 with random operands "
                                                             "(tools/fp64_peak.hip); this LDS-fed kernel sustains "
                                                             "2.1-2.25 GHz instead of 2.4 on real data "
                                                             "(tools/gemm_prof.hip, profiles/r01_run20_pipe_ab.txt)"},
            "corr_build_gbps": tim1[0]["corr_bytes"] / (corr_ms * 1e-3) / 1e9,
            "likelihood_checksum": float(np.sum(lkhs[args.warmup * world * nb:])),
        }
        if world == 1:
            # the boundary hands over HOST buffers: one cold call sequence including the device allocation (2 GiB
            # workspace), the upload of x / y over PCIe, the fit and the download of its scalars -- never `value`
            t0 = time.perf_counter()
            hcold = egx.GpHandle(x, y, mean=0, corr=0, device=local_rank, n_workspaces=1)
            t1 = time.perf_counter()
            hcold.finalize(cands[0])
            hcold.fitted_scalars()
            t2 = time.perf_counter()
            hcold.close()
            out["pcie_inclusive"] = {"create_alloc_upload_s": t1 - t0, "fit_and_download_s": t2 - t1,
                                     "fits_per_s_cold_handle": 1.0 / (t2 - t0),
                                     "note": "x, y (4 MiB) cross PCIe once per handle; every further fit on the handle "
                                             "moves (p + 2) n doubles back (0.4 MB)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(n, d)
            out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            rs = cpu_baseline_reference_shaped(n, d)
            if rs is not None:
                out["cpu_baseline_reference_shaped"] = rs
        print(json.dumps(out), flush=True)
    for g in gps:
        g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
